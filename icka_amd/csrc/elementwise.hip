// Element-wise / layout / reduction helpers of the ICKA hot path (gfx950).  All are HBM/L2-bound: every thread
// moves 16-byte chunks (8 bf16 or 4+4 f32), consecutive lanes touch consecutive chunks.
#include <math.h>

#include "common.h"

namespace {

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int grid_for(int64_t n, int per_block = 256, int cap = 4096) {
    int64_t g = (n + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
    const bf16x8 v = as_bf16x8(*reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(v[e]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<u32x4*>(p) = as_u32x4(o);
}

__global__ void cast_f2b_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n) {
    const int64_t nch = n >> 3;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + c * 8), b = *reinterpret_cast<const f32x4*>(src + c * 8 + 4);
        bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
        *reinterpret_cast<u32x4*>(dst + c * 8) = as_u32x4(o);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nch << 3) + threadIdx.x] = f2bf(src[(nch << 3) + threadIdx.x]);
}
// both 16-bit shadows of the f32 master weights in one pass: bf16 (dgrad / wgrad operands) and fp16 ("mixed16" forward)
__global__ void cast_f2bh_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, _Float16* __restrict__ dsth, int64_t n) {
    const int64_t nch = n >> 3;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(src + c * 8), b = *reinterpret_cast<const f32x4*>(src + c * 8 + 4);
        bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
        *reinterpret_cast<u32x4*>(dst + c * 8) = as_u32x4(o);
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (_Float16)fminf(fmaxf(a[e], -65504.f), 65504.f);
            h[4 + e] = (_Float16)fminf(fmaxf(b[e], -65504.f), 65504.f);
        }
        *reinterpret_cast<u32x4*>(dsth + c * 8) = __builtin_bit_cast(u32x4, h);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const float v = src[(nch << 3) + threadIdx.x];
        dst[(nch << 3) + threadIdx.x] = f2bf(v);
        dsth[(nch << 3) + threadIdx.x] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    }
}
// bf16 (kind 0) or f32 (kind 1) -> fp16, saturating: block inputs of the "mixed16" mode that arrive without an fp16 twin
__global__ void cast_to_f16_kernel(const void* __restrict__ src, int kind, _Float16* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = kind ? reinterpret_cast<const float*>(src)[i] : bf2f(reinterpret_cast<const bf16_t*>(src)[i]);
        dst[i] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);
    }
}
// 8 elements per thread and iteration (one 16-byte load, two 16-byte stores) when both pointers are 16-byte aligned: the
// data-parallel reducer casts 119 M gradients back per step (the scalar form below moves 2 + 4 bytes per instruction)
__global__ void cast_b2f_vec_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t nch = n >> 3;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = as_bf16x8(*reinterpret_cast<const u32x4*>(src + c * 8));
        *reinterpret_cast<f32x4*>(dst + c * 8) = f32x4{bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])};
        *reinterpret_cast<f32x4*>(dst + c * 8 + 4) = f32x4{bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7])};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nch << 3) + threadIdx.x] = bf2f(src[(nch << 3) + threadIdx.x]);
}
__global__ void cast_b2f_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = bf2f(src[i]);
}

__global__ void cast_pad_kernel(const float* __restrict__ src, int64_t lds, bf16_t* __restrict__ dst, int64_t ldd, int M,
                                int N) {
    const int64_t total = (int64_t)M * ldd;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ldd;
        const int c = (int)(i - r * ldd);
        dst[i] = f2bf(c < N ? src[r * lds + c] : 0.f);
    }
}

__global__ void additive_mask_kernel(const int64_t* __restrict__ mask, int64_t ld, float* __restrict__ out, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i - b * T;
    out[i] = (1.0f - (float)mask[(int64_t)b * ld + t]) * -10000.0f;
}

__global__ void dropout_kernel(const bf16_t* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y, int64_t ldy,
                               bf16_t* __restrict__ y2, int64_t ldy2, int M, int H, DropCfg d_) {
    const DropCfg d = drop_resolve(d_);
    const int cpr = H >> 3;
    const int64_t total = (int64_t)M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int c = (int)(i - row * cpr);
        float v[8];
        ld8(x + row * ldx + c * 8, v);
        const uint32_t base = (uint32_t)row * (uint32_t)H + c * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= drop_mul(d, base + e);
        st8(y + row * ldy + c * 8, v);
        if (y2) st8(y2 + row * ldy2 + c * 8, v);
    }
}

__global__ void dropout_mask_kernel(float* out, int64_t n, DropCfg d_) {
    const DropCfg d = drop_resolve(d_);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = drop_mul(d, (uint32_t)i);
}

// the multiplier (0 or 1/(1-p)) every attention kernel applies to element (row, key) of a probability matrix [rows, Skv]
__global__ void attn_dropout_mask_kernel(float* out, int64_t rows, int Skv, DropCfg d_) {
    const DropCfg d = drop_resolve(d_);
    const int64_t n = rows * Skv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / Skv;
        out[i] = d.thr ? drop_mul_key(d, (uint32_t)row * (uint32_t)Skv, (uint32_t)(i - row * Skv)) : d.scale;
    }
}

// layout 1: src f32 [B, C, R] -> dst bf16 [B, R, C]; one block per (b, 64-channel strip), transposed through LDS
__global__ __launch_bounds__(256) void regions_t_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int B,
                                                        int R, int C, _Float16* __restrict__ dsth = nullptr) {
    extern __shared__ float tile[];  // [64][R+1]
    const int strips = (C + 63) / 64;
    const int b = blockIdx.x / strips, c0 = (blockIdx.x % strips) * 64;
    const int nc = (C - c0) < 64 ? (C - c0) : 64;
    const float* s = src + ((int64_t)b * C + c0) * R;
    for (int i = threadIdx.x; i < nc * R; i += 256) {
        const int c = i / R, r = i - c * R;
        tile[c * (R + 1) + r] = s[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        if (c < nc) {
            const float v = tile[c * (R + 1) + r];
            dst[((int64_t)b * R + r) * C + c0 + c] = f2bf(v);
            if (dsth) dsth[((int64_t)b * R + r) * C + c0 + c] = (_Float16)fminf(fmaxf(v, -65504.f), 65504.f);   // "mixed16" twin
        }
    }
}

constexpr int CS_GROUPS = 256;
// partial[group][N] = sum over the group's rows; block = 64 chunk-lanes x 4 row-lanes
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, int64_t ldx, float* __restrict__ partial,
                                                     int M, int N, int rows_per_group, int vec) {
    __shared__ float red[4][64][8];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + cx) * 8;
    const int r0 = blockIdx.y * rows_per_group;
    int r1 = r0 + rows_per_group; r1 = r1 > M ? M : r1;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (col < N) {
        const int nv = (N - col) < 8 ? (N - col) : 8;
        for (int r = r0 + ry; r < r1; r += 4) {
            const bf16_t* p = x + (int64_t)r * ldx + col;
            if (nv == 8 && vec) {
                float v[8];
                ld8(p, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[e];
            } else {
                for (int e = 0; e < nv; ++e) acc[e] += bf2f(p[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ry][cx][e] = acc[e];
    __syncthreads();
    if (ry == 0 && col < N) {
        for (int e = 0; e < 8 && col + e < N; ++e)
            partial[(int64_t)blockIdx.y * N + col + e] = red[0][cx][e] + red[1][cx][e] + red[2][cx][e] + red[3][cx][e];
    }
}
__global__ __launch_bounds__(1024) void colsum_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int groups, int N, int accumulate) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (c < N)
        for (int g = sy; g < groups; g += 16) s += partial[(int64_t)g * N + c];
    red[sy][cx] = s;
    __syncthreads();
    if (sy == 0 && c < N) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

__global__ void gate_bwd_kernel(const bf16_t* __restrict__ dout, int64_t lddout, const bf16_t* __restrict__ g,
                                const bf16_t* __restrict__ cross, int64_t ldcross, const bf16_t* __restrict__ dci,
                                int64_t lddci, bf16_t* __restrict__ du, bf16_t* __restrict__ dcross, int64_t lddc, int M,
                                int H) {
    const int cpr = H >> 3;
    const int64_t total = (int64_t)M * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int c = (int)(i - row * cpr) * 8;
        float d[8], gg[8], cr[8], u[8], dc[8];
        ld8(dout + row * lddout + c, d);
        ld8(g + row * H + c, gg);
        ld8(cross + row * ldcross + c, cr);
#pragma unroll
        for (int e = 0; e < 8; ++e) { u[e] = d[e] * cr[e] * gg[e] * (1.f - gg[e]); dc[e] = d[e] * gg[e]; }
        if (dci) {
            float t[8];
            ld8(dci + row * lddci + c, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) dc[e] += t[e];
        }
        st8(du + row * H + c, u);
        st8(dcross + row * lddc + c, dc);
    }
}

__global__ void add_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ c, int64_t n) {
    const int64_t nch = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nch; i += (int64_t)gridDim.x * blockDim.x) {
        float x[8], y[8];
        ld8(a + i * 8, x); ld8(b + i * 8, y);
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] += y[e];
        st8(c + i * 8, x);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t j = (nch << 3) + threadIdx.x;
        c[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
    }
}

// dz = dg * gelu'(z): backward of BertIntermediate's activation when the module is called on its own (inside BertLayer
// it is the DGELU epilogue of the d(ffn-down) GEMM)
__global__ void dgelu_kernel(const bf16_t* __restrict__ dg, const bf16_t* __restrict__ z, bf16_t* __restrict__ dz, int64_t n) {
    const int64_t nch = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nch; i += (int64_t)gridDim.x * blockDim.x) {
        float g[8], t[8];
        ld8(dg + i * 8, g); ld8(z + i * 8, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= dgelu_f(t[e]);
        st8(dz + i * 8, g);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t j = (nch << 3) + threadIdx.x;
        dz[j] = f2bf(bf2f(dg[j]) * dgelu_f(bf2f(z[j])));
    }
}

// dx = dy * (1 - y^2): backward of y = tanh(.) (the prompt mapping networks' activation)
__global__ void tanh_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, bf16_t* __restrict__ dx, int64_t n) {
    const int64_t nch = n >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nch; i += (int64_t)gridDim.x * blockDim.x) {
        float g[8], t[8];
        ld8(dy + i * 8, g); ld8(y + i * 8, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= 1.f - t[e] * t[e];
        st8(dx + i * 8, g);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        const int64_t j = (nch << 3) + threadIdx.x;
        const float t = bf2f(y[j]);
        dx[j] = f2bf(bf2f(dy[j]) * (1.f - t * t));
    }
}

__global__ __launch_bounds__(256) void token_ce_kernel(const float* __restrict__ logits, int64_t ld,
                                                       const int64_t* __restrict__ labels, const int64_t* __restrict__ mask,
                                                       float* loss_sum, float* count, bf16_t* __restrict__ dl, int64_t ldd,
                                                       int M, int C) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    float loss = 0.f, cnt = 0.f;
    if (row < M) {
        const float* p = logits + (int64_t)row * ld;
        const bool valid = mask[row] != 0;
        const int64_t y = labels[row];
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        const float lse = mx + logf(se);
        bf16_t* d = dl + (int64_t)row * ldd;
        for (int c = 0; c < (int)ldd; ++c) {
            float gval = 0.f;
            if (valid && c < C) gval = expf(p[c] - lse) - (c == (int)y ? 1.f : 0.f);
            d[c] = f2bf(gval);
        }
        if (valid && y >= 0 && y < C) { loss = lse - p[y]; cnt = 1.f; }
    }
    loss = wave_sum(loss);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0 && cnt > 0.f) { atomicAdd(loss_sum, loss); atomicAdd(count, cnt); }
}

// Token-CE without pre-zeroed accumulators and without atomics: every 256-row block leaves its (loss, count) partial in
// part[block][2]; ce_finalize_kernel sums them in a fixed order and writes stats = {sum, count, mean} (this second launch
// replaces the scalar-ratio launch of the atomic version, the accumulator fill disappears).
__global__ __launch_bounds__(256) void token_ce_part_kernel(const float* __restrict__ logits, int64_t ld,
                                                            const int64_t* __restrict__ labels,
                                                            const int64_t* __restrict__ mask, float* __restrict__ part,
                                                            void* __restrict__ dl_, int64_t ldd, int dl_f32, int M, int C) {
    __shared__ float red[2][4];
    const int row = blockIdx.x * 256 + threadIdx.x;
    float loss = 0.f, cnt = 0.f;
    if (row < M) {
        const float* p = logits + (int64_t)row * ld;
        const bool valid = mask[row] != 0;
        const int64_t y = labels[row];
        float mx = -INFINITY;
        for (int c = 0; c < C; ++c) mx = fmaxf(mx, p[c]);
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(p[c] - mx);
        const float lse = mx + logf(se);
        for (int c = 0; c < (int)ldd; ++c) {
            float gval = 0.f;
            if (valid && c < C) gval = expf(p[c] - lse) - (c == (int)y ? 1.f : 0.f);
            if (dl_f32) reinterpret_cast<float*>(dl_)[(int64_t)row * ldd + c] = gval;
            else reinterpret_cast<bf16_t*>(dl_)[(int64_t)row * ldd + c] = f2bf(gval);
        }
        if (valid && y >= 0 && y < C) { loss = lse - p[y]; cnt = 1.f; }
    }
    loss = wave_sum(loss);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = loss; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        part[2 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}
__global__ __launch_bounds__(64) void ce_finalize_kernel(const float* __restrict__ part, int nblk, float* __restrict__ stats) {
    float l = 0.f, n = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) { l += part[2 * b]; n += part[2 * b + 1]; }
    l = wave_sum(l);
    n = wave_sum(n);
    if (threadIdx.x == 0) { stats[0] = l; stats[1] = n; stats[2] = l / fmaxf(n, 1.f); }
}

// 16-byte stores, enough blocks to cover the HBM channels (the 94 MB word-embedding gradient table is cleared once per
// accumulation cycle before the atomic scatter: at::fill_ took ~70 us for it, ~1.3 TB/s)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int64_t n) {
    const int64_t nch = n >> 2;
    const u32x4 z = {0u, 0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nch; i += (int64_t)gridDim.x * 256)
        __builtin_nontemporal_store(z, reinterpret_cast<u32x4*>(p) + i);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(nch << 2) + threadIdx.x] = 0.f;
}

__global__ void scale_ratio_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, const float* num,
                                   const float* den, int64_t n) {
    const float s = (num ? num[0] : 1.f) / (den ? fmaxf(den[0], 1.f) : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = f2bf(bf2f(x[i]) * s);
}
__global__ void bump_nonce_kernel(uint32_t* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t a = p[0] + 0x9E3779B9u;
        p[0] = a;
        p[1] = icka_hash(a, 0x85EBCA6Bu, p[1]);
    }
}
__global__ void scalar_ratio_kernel(float* out, const float* num, const float* den) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = num[0] / fmaxf(den[0], 1.f);
}

// Several device-to-device copies in ONE launch (the per-call input refresh of a captured step: ids, masks, labels, region
// features -- seven tensors of 32 KB .. 9.4 MB at c2).  blockIdx.y = copy, blockIdx.x strides over its 16-byte words; a
// copy that is not 16-byte sized / aligned goes byte by byte (small tensors only).
constexpr int COPY_MAX = 8;
struct CopyMany {
    const char* src[COPY_MAX];
    char* dst[COPY_MAX];
    long long bytes[COPY_MAX];
};
__global__ __launch_bounds__(256) void copy_many_kernel(const CopyMany t) {
    const int e = blockIdx.y;
    const long long nb = t.bytes[e];
    const char* s = t.src[e];
    char* d = t.dst[e];
    const bool vec = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | (uintptr_t)nb) & 15) == 0;
    const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nth = (long long)gridDim.x * 256;
    if (vec) {
        const long long nv = nb >> 4;
        for (long long i = tid; i < nv; i += nth)
            reinterpret_cast<u32x4*>(d)[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(s) + i);
    } else {
        for (long long i = tid; i < nb; i += nth) d[i] = s[i];
    }
}

}  // namespace

const uint32_t* g_icka_nonce = nullptr;
int g_icka_reserved_cus = 0;

extern "C" int icka_abi_version(void) { return ICKA_ABI_VERSION; }
extern "C" int icka_set_dropout_nonce(const uint32_t* device_words) {
    g_icka_nonce = device_words;
    return 0;
}
extern "C" int icka_bump_dropout_nonce(uint32_t* device_words, void* stream) {
    if (!device_words) return ICKA_E_ARG;
    hipLaunchKernelGGL(bump_nonce_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, device_words);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" const char* icka_build_arch(void) { return "gfx950"; }

extern "C" int icka_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream) {
    if (n < 0 || n > COPY_MAX || (n > 0 && (!src || !dst || !bytes))) return ICKA_E_ARG;
    if (n == 0) return 0;
    CopyMany t{};
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        if (bytes[i] < 0 || (bytes[i] > 0 && (!src[i] || !dst[i]))) return ICKA_E_ARG;
        t.src[i] = (const char*)src[i];
        t.dst[i] = (char*)dst[i];
        t.bytes[i] = bytes[i];
        most = bytes[i] > most ? bytes[i] : most;
    }
    long long blocks = (most / 16 + 256 * 4 - 1) / (256 * 4);     // ~4 x 16 bytes per thread for the largest copy
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, t);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (!al16(src) || !al16(dst)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(cast_f2b_kernel, dim3(grid_for((n + 7) / 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_cast_f32_to_bf16_f16(const float* src, void* dst_bf16, void* dst_f16, int64_t n, void* stream) {
    if (!src || !dst_bf16 || !dst_f16) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (!al16(src) || !al16(dst_bf16) || !al16(dst_f16)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(cast_f2bh_kernel, dim3(grid_for((n + 7) / 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, src,
                       (bf16_t*)dst_bf16, (_Float16*)dst_f16, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_cast_to_f16(const void* src, int32_t src_is_f32, void* dst, int64_t n, void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(cast_to_f16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (int)src_is_f32,
                       (_Float16*)dst, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (al16(src) && al16(dst))
        hipLaunchKernelGGL(cast_b2f_vec_kernel, dim3(grid_for((n + 7) / 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)src, dst, n);
    else
        hipLaunchKernelGGL(cast_b2f_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_cast_pad_f32_to_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int32_t M, int32_t N,
                                         void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (M <= 0 || N <= 0 || ldd < N || lds < N) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(cast_pad_kernel, dim3(grid_for((int64_t)M * ldd)), dim3(256), 0, (hipStream_t)stream, src, lds,
                       (bf16_t*)dst, ldd, M, N);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_additive_mask(const int64_t* mask, int64_t ld, float* out, int32_t B, int32_t T, void* stream) {
    if (!mask || !out) return ICKA_E_ARG;
    if (B <= 0 || T <= 0 || ld < T) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(additive_mask_kernel, dim3((B * T + 255) / 256), dim3(256), 0, (hipStream_t)stream, mask, ld, out, B, T);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dropout(const void* x, int64_t ldx, void* y, int64_t ldy, void* y2, int64_t ldy2, int32_t M,
                            int32_t H, float p_drop, uint64_t seed, void* stream) {
    if (!x || !y) return ICKA_E_ARG;
    if (M <= 0 || H <= 0 || H % 8) return ICKA_E_SHAPE;
    if (ldx % 8 || ldy % 8 || (y2 && ldy2 % 8) || !al16(x) || !al16(y) || (y2 && !al16(y2))) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((int64_t)M * (H / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, ldx, (bf16_t*)y, ldy, (bf16_t*)y2, ldy2, M, H, make_drop(p_drop, seed));
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dropout_mask(float* out, int64_t n, float p_drop, uint64_t seed, void* stream) {
    if (!out) return ICKA_E_ARG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, n,
                       make_drop(p_drop, seed));
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_attn_dropout_mask(float* out, int64_t rows, int32_t Skv, float p_drop, uint64_t seed, void* stream) {
    if (!out) return ICKA_E_ARG;
    if (rows <= 0 || Skv <= 0) return 0;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(grid_for(rows * Skv)), dim3(256), 0, (hipStream_t)stream, out, rows, Skv,
                       make_drop(p_drop, seed));
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_regions_to_tokens(const float* src, void* dst, int32_t B, int32_t R, int32_t C, int32_t layout,
                                      void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (B <= 0 || R <= 0 || C <= 0 || R > 256) return ICKA_E_SHAPE;
    if (layout == 0) return icka_cast_f32_to_bf16(src, dst, (int64_t)B * R * C, stream);
    if (layout != 1) return ICKA_E_ARG;
    hipLaunchKernelGGL(regions_t_kernel, dim3(B * ((C + 63) / 64)), dim3(256), 64 * (R + 1) * sizeof(float),
                       (hipStream_t)stream, src, (bf16_t*)dst, B, R, C);
    ICKA_CHECK_LAUNCH();
    return 0;
}
/* "mixed16": region tokens in both 16-bit types in one pass over the f32 features (the fp16 copy is the forward operand of
   the region projection, the bf16 copy the operand of its weight gradient). */
extern "C" int icka_regions_to_tokens_h(const float* src, void* dst_bf16, void* dst_f16, int32_t B, int32_t R, int32_t C,
                                        int32_t layout, void* stream) {
    if (!src || !dst_bf16 || !dst_f16) return ICKA_E_ARG;
    if (B <= 0 || R <= 0 || C <= 0 || R > 256) return ICKA_E_SHAPE;
    if (layout == 0) return icka_cast_f32_to_bf16_f16(src, dst_bf16, dst_f16, (int64_t)B * R * C, stream);
    if (layout != 1) return ICKA_E_ARG;
    hipLaunchKernelGGL(regions_t_kernel, dim3(B * ((C + 63) / 64)), dim3(256), 64 * (R + 1) * sizeof(float),
                       (hipStream_t)stream, src, (bf16_t*)dst_bf16, B, R, C, (_Float16*)dst_f16);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int64_t icka_colsum_workspace_floats(int32_t N) { return (int64_t)CS_GROUPS * N; }
extern "C" int icka_colsum(const void* x, int64_t ldx, float* out, float* partials, int32_t M, int32_t N,
                           int32_t accumulate, void* stream) {
    if (!x || !out || !partials) return ICKA_E_ARG;
    if (M <= 0 || N <= 0) return ICKA_E_SHAPE;
    const int vec = (ldx % 8 == 0) && al16(x);
    int groups = (M + 15) / 16; groups = groups > CS_GROUPS ? CS_GROUPS : groups;
    const int rpg = (M + groups - 1) / groups;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, groups), dim3(256), 0, st, (const bf16_t*)x, ldx, partials, M,
                       N, rpg, vec);
    ICKA_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, partials, out, groups, N, accumulate);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_gate_bwd(const void* dout, int64_t lddout, const void* g, const void* cross, int64_t ldcross,
                             const void* dcross_in, int64_t lddci, void* du, void* dcross, int64_t lddc, int32_t M,
                             int32_t H, void* stream) {
    if (!dout || !g || !cross || !du || !dcross) return ICKA_E_ARG;
    if (M <= 0 || H <= 0 || H % 8) return ICKA_E_SHAPE;
    if (lddout % 8 || ldcross % 8 || lddc % 8 || (dcross_in && lddci % 8)) return ICKA_E_ALIGN;
    if (!al16(dout) || !al16(g) || !al16(cross) || !al16(du) || !al16(dcross) || (dcross_in && !al16(dcross_in)))
        return ICKA_E_ALIGN;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for((int64_t)M * (H / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dout, lddout, (const bf16_t*)g, (const bf16_t*)cross, ldcross,
                       (const bf16_t*)dcross_in, lddci, (bf16_t*)du, (bf16_t*)dcross, lddc, M, H);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_add_bf16(const void* a, const void* b, void* c, int64_t n, void* stream) {
    if (!a || !b || !c) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (!al16(a) || !al16(b) || !al16(c)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                       (const bf16_t*)b, (bf16_t*)c, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dgelu_bf16(const void* dg, const void* z, void* dz, int64_t n, void* stream) {
    if (!dg || !z || !dz) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (!al16(dg) || !al16(z) || !al16(dz)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(dgelu_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dg,
                       (const bf16_t*)z, (bf16_t*)dz, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_tanh_bwd(const void* dy, const void* y, void* dx, int64_t n, void* stream) {
    if (!dy || !y || !dx) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (!al16(dy) || !al16(y) || !al16(dx)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy,
                       (const bf16_t*)y, (bf16_t*)dx, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_token_ce(const float* logits, int64_t ld, const int64_t* labels, const int64_t* mask,
                             float* loss_sum, float* count, void* dlogits, int64_t ldd, int32_t M, int32_t C,
                             void* stream) {
    if (!logits || !labels || !mask || !loss_sum || !count || !dlogits) return ICKA_E_ARG;
    if (M <= 0 || C <= 0 || ldd < C || ld < C) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(token_ce_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, ld, labels,
                       mask, loss_sum, count, (bf16_t*)dlogits, ldd, M, C);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int64_t icka_token_ce_workspace_floats(int32_t M) { return 2 * (int64_t)((M + 255) / 256); }
extern "C" int icka_token_ce_fused(const float* logits, int64_t ld, const int64_t* labels, const int64_t* mask, float* stats,
                                   float* partials, void* dlogits, int64_t ldd, int32_t dl_is_f32, int32_t M, int32_t C,
                                   void* stream) {
    if (!logits || !labels || !mask || !stats || !partials || !dlogits) return ICKA_E_ARG;
    if (M <= 0 || C <= 0 || ldd < C || ld < C) return ICKA_E_SHAPE;
    const int nblk = (M + 255) / 256;
    hipLaunchKernelGGL(token_ce_part_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, logits, ld, labels, mask,
                       partials, dlogits, ldd, dl_is_f32, M, C);
    ICKA_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partials, nblk, stats);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_zero_f32(float* p, int64_t n, void* stream) {
    if (!p) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if (!al16(p)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(zero_f32_kernel, dim3(grid_for((n + 3) / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, p, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_scale_by_ratio(const void* x, void* y, const float* num, const float* den, int64_t n,
                                   void* stream) {
    if (!x || !y) return ICKA_E_ARG;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scale_ratio_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                       (bf16_t*)y, num, den, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_scalar_ratio(float* out, const float* num, const float* den, void* stream) {
    if (!out || !num || !den) return ICKA_E_ARG;
    hipLaunchKernelGGL(scalar_ratio_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, num, den);
    ICKA_CHECK_LAUNCH();
    return 0;
}
