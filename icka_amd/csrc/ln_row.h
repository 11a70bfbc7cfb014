// ln_row.h -- the forward row body of the fused bias + dropout + residual + LayerNorm (BertSelfOutput.forward
// Cross_Modal_Interaction_Module.py:561-565, BertOutput.forward :532-536, BertLayerNorm.forward :518-522), shared by the
// stand-alone row kernel (layernorm.hip: ln_fwd_kernel) and by the GEMM kernel that finishes its own rows (gemm.hip:
// gemm_ln_kernel): ONE definition of the arithmetic, so the two paths give bitwise the same y / twin / xhat / rstd.
#pragma once
#include "common.h"

namespace {

constexpr int MAX_CH = 4;           // chunks of 8 per lane -> H <= 2048

struct LnFwdArgs {
    const void* x; int64_t ldx; int x_f32; const float* bias; const void* res; int64_t ldr; int r_f32;
    const float* gamma; const float* beta;
    bf16_t* y; int64_t ldy; bf16_t* y2; int64_t ldy2; void* yf; bf16_t* xhat; float* rstd;
    int M, H; float eps; DropCfg drop;
    int yf_f16;   // the twin output ``yf`` is fp16 (the "mixed16" forward operand + residual) instead of f32
};

__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
    const bf16x8 v = as_bf16x8(*reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(v[e]);
}
__device__ __forceinline__ void load8f(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void load8h(const _Float16* p, float (&o)[8]) {
    const f16x8 v = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
// row element loads in either precision: kind 0 = bf16, 1 = f32 (GEMM outputs, the f32 residual twin), 2 = fp16 (the
// residual stream of the "mixed16" mode)
__device__ __forceinline__ void load8x(const void* base, int kind, int64_t off, float (&o)[8]) {
    if (kind == 1) load8f(reinterpret_cast<const float*>(base) + off, o);
    else if (kind == 2) load8h(reinterpret_cast<const _Float16*>(base) + off, o);
    else load8(reinterpret_cast<const bf16_t*>(base) + off, o);
}
__device__ __forceinline__ void store8f(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<u32x4*>(p) = as_u32x4(o);
}

__device__ __forceinline__ void store8h(_Float16* p, const float (&v)[8]) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)fminf(fmaxf(v[e], -65504.f), 65504.f);
    *reinterpret_cast<u32x4*>(p) = __builtin_bit_cast(u32x4, o);
}
// the twin copy of a LayerNorm output: f32, or fp16 in the "mixed16" mode
__device__ __forceinline__ void store8t(void* base, int is_f16, int64_t off, const float (&v)[8]) {
    if (is_f16) store8h(reinterpret_cast<_Float16*>(base) + off, v);
    else store8f(reinterpret_cast<float*>(base) + off, v);
}

// One wave per row at a time; a wave that owns several rows (grid < M / 4) issues the loads of its
// NEXT row before it reduces and stores the current one, so that a CU's read and write streams overlap instead of the whole
// chip reading, then reducing, then writing in step.
template <int NCH>
__device__ __forceinline__ void ln_load_raw(const LnFwdArgs& a, int row, int lane, int nchunk, float (&x)[NCH][8], float (&r)[NCH][8]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            load8x(a.x, a.x_f32, (int64_t)row * a.ldx + c * 8, x[i]);
            if (a.res) load8x(a.res, a.r_f32, (int64_t)row * a.ldr + c * 8, r[i]);
        }
    }
}

// x loads that bypass this CU's L1 (sc1): the row was written by OTHER CUs of the same launch (gemm_ln_kernel: write-through
// sc1 stores, s_waitcnt vmcnt(0), then the stripe's arrival counter) -- L2 / memory serve it, whatever this CU's L1 still holds of
// the buffer from an earlier launch.  Raw buffer loads with the sc1 cache-policy bit (aux 16): 16-byte loads the compiler
// tracks itself (waits, hazards) -- inline-asm loads were tried first: hipcc copies their result registers before the wait.
constexpr int ICKA_CPOL_SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ln_x_rsrc(const LnFwdArgs& a) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ void load8f_sc1(__amdgpu_buffer_rsrc_t r, int byte_off, float (&o)[8]) {
    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, ICKA_CPOL_SC1);
    const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off + 16, 0, ICKA_CPOL_SC1);
    o[0] = __uint_as_float(lo[0]); o[1] = __uint_as_float(lo[1]); o[2] = __uint_as_float(lo[2]); o[3] = __uint_as_float(lo[3]);
    o[4] = __uint_as_float(hi[0]); o[5] = __uint_as_float(hi[1]); o[6] = __uint_as_float(hi[2]); o[7] = __uint_as_float(hi[3]);
}
// ONE row from its raw operands to its stores: s = the dense output (x), rr = the residual (ignored without a.res); on return s holds
// the pre-normalisation row.  The single definition of the arithmetic of every forward path.
template <int NCH>
__device__ __forceinline__ void ln_row_finish(const LnFwdArgs& a, int row, int lane, float (&s)[NCH][8], const float (&rr)[NCH][8]) {
    const int nchunk = a.H >> 3;
    const float inv_h = 1.f / (float)a.H;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            if (a.bias) {
                float b[8];
                load8f(a.bias + c * 8, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[i][e] += b[e];
            }
            if (a.drop.thr) {
                const uint32_t base = (uint32_t)row * (uint32_t)a.H + c * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) s[i][e] *= drop_mul(a.drop, base + e);
            }
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[i][e] += rr[i][e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += s[i][e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) s[i][e] = 0.f;
        }
    }
    const float mean = wave_sum(sum) * inv_h;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (lane + 64 * i < nchunk) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = s[i][e] - mean; sq += d * d; }
        }
    const float var = wave_sum(sq) * inv_h;
    const float rstd = 1.f / sqrtf(var + a.eps);
    if (lane == 0 && a.rstd) a.rstd[row] = rstd;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            float g[8], b[8], xh[8], o[8];
            load8f(a.gamma + c * 8, g);
            load8f(a.beta + c * 8, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) { xh[e] = (s[i][e] - mean) * rstd; o[e] = g[e] * xh[e] + b[e]; }
            store8(a.y + (int64_t)row * a.ldy + c * 8, o);
            if (a.y2) store8(a.y2 + (int64_t)row * a.ldy2 + c * 8, o);
            if (a.yf) store8t(a.yf, a.yf_f16, (int64_t)row * a.H + c * 8, o);
            if (a.xhat) store8(a.xhat + (int64_t)row * a.H + c * 8, xh);
        }
    }
}

// Rows row0, row0 + step, ... < row_end of ONE wave (the stand-alone row kernel).  ``a.drop`` is already resolved (drop_resolve).
template <int NCH>
__device__ __forceinline__ void ln_fwd_rows(const LnFwdArgs& a, int row0, int step, int row_end, int lane) {
    const int nchunk = a.H >> 3;
    float s[NCH][8], rr[NCH][8];
    if (row0 < row_end) ln_load_raw<NCH>(a, row0, lane, nchunk, s, rr);
    for (int row = row0; row < row_end; row += step) {
        float nx[NCH][8], nr[NCH][8];
        const int nrow = row + step;
        if (nrow < row_end) ln_load_raw<NCH>(a, nrow, lane, nchunk, nx, nr);   // in flight while this row is reduced and stored
        ln_row_finish<NCH>(a, row, lane, s, rr);
        if (nrow < row_end) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[i][e] = nx[i][e]; rr[i][e] = nr[i][e]; }
        }
    }
}

// The LayerNorm phase of gemm_ln_kernel, per wave: NR consecutive rows whose RESIDUAL operands were loaded before the GEMM
// started (ln_prefetch_res: they do not depend on it, and after the seam the whole chip would ask HBM for them at once) and
// whose x arrives from the other blocks of the stripe: all NR rows' x loads (sc1: L2 / memory, not this CU's L1) are issued
// at once, then the rows are finished one after the other.
template <int NCH, int NR>
__device__ __forceinline__ void ln_prefetch_res(const LnFwdArgs& a, int row0, int lane, float (&rr)[NR][NCH][8]) {
    const int nchunk = a.H >> 3;
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (a.res && c < nchunk) load8x(a.res, a.r_f32, (int64_t)(row0 + r) * a.ldr + c * 8, rr[r][i]);
        }
}
template <int NCH, int NR>
__device__ __forceinline__ void ln_fwd_rows_handoff(const LnFwdArgs& a, int row0, int lane, const float (&rr)[NR][NCH][8]) {
    const int nchunk = a.H >> 3;
    __amdgpu_buffer_rsrc_t xr = ln_x_rsrc(a);
    float s[NR][NCH][8];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) load8f_sc1(xr, (int)(((int64_t)(row0 + r) * a.ldx + c * 8) * 4), s[r][i]);      // (host: M * ldx * 4 < 2^31)
        }
#pragma unroll
    for (int r = 0; r < NR; ++r) ln_row_finish<NCH>(a, row0 + r, lane, s[r], rr[r]);
}

}  // namespace
