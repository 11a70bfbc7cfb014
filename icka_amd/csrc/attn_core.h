// Shared by attention.hip and gemm.hip: the LDS tile image of a head's Q / K / V rows, the MFMA fragment reads, the argument
// block of the attention launches, and the WHOLE-HEAD forward (every score of the head in registers) as a device function --
// the body of attn_fwd_small_kernel, which gemm_qkv_attn_kernel (gemm.hip) runs in the epilogue of the QKV projection on tiles
// that never left the LDS.  One definition, so that the two launches and the fused one are the same arithmetic bit for bit.
#pragma once
#include <math.h>

#include "common.h"

namespace {

constexpr int HD = 64;           // head size
constexpr int TILE = 64;         // rows per LDS tile
constexpr int TILE_B = TILE * HD * 2;

// [64 rows][64 d] bf16 tile image serving both row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16):
// 8-row x 32-col sub-tiles of 512 B, 16-B chunk XORed with (row>>2)&3 (cdna guide T10, image (a)).
__device__ __forceinline__ uint32_t off_t(int row, int ch) {
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// Stage rows [row0, row0+64) of a token-major matrix (this head's 64 columns) into an LDS tile; rows >= nrows -> 0.
__device__ __forceinline__ void stage_tile(char* lds, const bf16_t* base, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, r = q >> 3, c = q & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < nrows) v = *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + r) * ld + c * 8);
        *reinterpret_cast<u32x4*>(lds + off_t(r, c)) = v;
    }
}

// row-read fragment: element e = tile[rbase + (l&15)][32*ks + 8*(l>>4) + e]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int rbase, int ks, int lane) {
    return lds_read_b128(tile, off_t(rbase + (lane & 15), 4 * ks + (lane >> 4)));
}
// transposed fragment for the "accumulator as B operand" k-slot order: lane l, element e (0..7) =
//   tile[row = 32*ks + 16*(e>>2) + 4*(l>>4) + (e&3)][col = 16*dt + (l&15)]
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int dt, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 32 * ks + 4 * g + q, ch = 2 * dt + (p >> 1);
    const bf16x4 lo = lds_read_tr(tile, off_t(row, ch) + 8 * (p & 1));
    const bf16x4 hi = lds_read_tr(tile, off_t(row + 16, ch) + 8 * (p & 1));
    return join8(lo, hi);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return o;
}

struct AttnArgs {
    const bf16_t* Q; int64_t ldq; const bf16_t* K; int64_t ldk; const bf16_t* V; int64_t ldv;
    const float* mask; const bf16_t* O; int64_t ldo; const bf16_t* dO; int64_t lddo;
    bf16_t* Ow; float* lse; float* delta;
    _Float16* Ow16;   // optional fp16 copy of the context (same leading dimension): "mixed16" forward operand of out-proj
    bf16_t* dQ; int64_t lddq; bf16_t* dK; int64_t lddk; bf16_t* dV; int64_t lddv;
    int B, h, Sq, Skv; float scale; DropCfg drop;
    // optional keep bits of the attention-probability dropout, written by the whole-head forward and read by the whole-head
    // backward instead of hashing again (the hash is about half of the backward's vector work per element): per (batch*head,
    // query, lane group g = (key % 16) / 4) WPL = ceil(Skv / 128) words, bit (key / 16) * 4 + key % 4 of the group's words
    uint32_t* keepbits;
#ifdef ICKA_ATTN_STAMP
    unsigned long long* stamp;   // diagnostic build: [block][wave][16] s_memtime / s_memrealtime stamps (tools/attn_stamp.py)
#endif
};
#ifdef ICKA_ATTN_STAMP
#define ATTN_STAMP(i) do { unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); if (a.stamp && lane == 0) a.stamp[((int64_t)blockIdx.x * 4 + wave) * 16 + (i)] = t__; } while (0)
#define ATTN_RSTAMP(i) do { unsigned long long t__; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory"); if (a.stamp && lane == 0) a.stamp[((int64_t)blockIdx.x * 4 + wave) * 16 + (i)] = t__; } while (0)
#else
#define ATTN_STAMP(i)
#define ATTN_RSTAMP(i)
#endif

// FP8 (BASELINE config c5): QK^T and PV on the fp8 matrix cores (v_mfma_f32_16x16x32_fp8_fp8, OCP e4m3 on gfx950).
// The bf16 fragments are converted in registers after the same LDS reads -- an fp8 fragment has the same lane layout
// as the bf16 one (8 consecutive k per lane).  P is scaled by 2^8 before the conversion (probabilities live far below
// e4m3's normal range: min normal 2^-6) and the accumulators are scaled back; softmax statistics stay fp32.
__device__ __forceinline__ long fp8x8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a0, a1, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(a2, a3, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(a4, a5, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(a6, a7, hi, true);
    return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ long fp8x8(const bf16x8& v) {
    return fp8x8(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3]), bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]));
}
__device__ __forceinline__ f32x4 mfma16_fp8(long a, long b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, c, 0, 0, 0);
}
constexpr float FP8_P_SCALE = 256.f;

// keep decisions of the four in-lane keys 16 kt + 4 g .. + 3 as a nibble (bit r = key r kept): the two pair hashes of
// drop_pair_x, x = (row * Skv / ... pair index) * C0 + s0 already formed by the caller for the first pair
template <bool DROP>
__device__ __forceinline__ uint32_t drop_nibble_x(const DropCfg& d, uint32_t x) {
    if (!DROP) return 0xFu;
    const uint32_t h0 = icka_hash_tail(x, d.s1), h1 = icka_hash_tail(x + ICKA_HASH_C0, d.s1), t = d.thr >> 16;
    return ((h0 & 0xffffu) >= t ? 1u : 0u) | ((h0 >> 16) >= t ? 2u : 0u) | ((h1 & 0xffffu) >= t ? 4u : 0u) | ((h1 >> 16) >= t ? 8u : 0u);
}
__host__ __device__ __forceinline__ int keep_wpl(int Skv) { return (Skv + 127) >> 7; }

// Forward of one wave's QT 16-query sub-tiles against the head's 16 * KT keys.  sQ / sK / sV: off_t images of the head's
// rows (at least 16 * (wq + QT) query rows, 16 * KT key rows); wq = index of the wave's first 16-query sub-tile inside sQ,
// q0 = query index (inside the sample) of sQ's row 0; bh = batch * heads + head (dropout counter, lse / keep-bit rows).
// a.drop must already be resolved (drop_resolve).  Writes the context rows (a.Ow / a.Ow16), a.lse and the keep bits.
template <int QT, int KT, bool DROP, bool FP8, bool KB>
__device__ __forceinline__ void attn_fwd_whole_head(const AttnArgs& a, const char* sQ, const char* sK, const char* sV, int wq,
                                                    int q0, int bh, int b, int head, int lane) {
    const int g = lane >> 4, i15 = lane & 15;
    const float* mb = a.mask + (int64_t)b * a.Skv;
    bf16x8 qf[QT][2];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qi][ks] = frag_row(sQ, 16 * (wq + qi), ks, lane);
    f32x4 s[QT][KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const bf16x8 k0 = frag_row(sK, 16 * kt, 0, lane), k1 = frag_row(sK, 16 * kt, 1, lane);
        f32x4 mk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            const float mv = mb[key < a.Skv ? key : a.Skv - 1];   // clamped load + select: no divergent branch
            mk[r] = key < a.Skv ? mv : -INFINITY;
        }
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            f32x4 t;
            if constexpr (FP8) {
                t = mfma16_fp8(fp8x8(k0), fp8x8(qf[qi][0]), f32x4{0.f, 0.f, 0.f, 0.f});
                t = mfma16_fp8(fp8x8(k1), fp8x8(qf[qi][1]), t);
            } else {
                t = mfma16(k0, qf[qi][0], f32x4{0.f, 0.f, 0.f, 0.f});
                t = mfma16(k1, qf[qi][1], t);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = t[r] * a.scale + mk[r];
            s[qi][kt] = t;
        }
    }
    bf16x8 pf[QT][KT / 2];
    long pf8[QT][KT / 2];
    float inv[QT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        const int q = q0 + 16 * (wq + qi) + i15;
        const uint32_t idx_row = ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;
        const uint32_t hx = (idx_row + 2u * (uint32_t)g) * ICKA_HASH_C0 + a.drop.s0;   // + (8 kt + r / 2) * C0 per PAIR of keys
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[qi][kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float psum = 0.f;
        uint32_t kbw[KB ? (KT * 4 + 31) / 32 : 1];
        if constexpr (KB) {
#pragma unroll
            for (int w = 0; w < (KT * 4 + 31) / 32; ++w) kbw[w] = 0u;
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            float dm4[4];
            if constexpr (KB) {   // the same two pair hashes, kept as bits for the backward as well
                const uint32_t nib = drop_nibble_x<true>(a.drop, hx + (uint32_t)(8 * kt) * ICKA_HASH_C0);
                kbw[(kt * 4) >> 5] |= nib << ((kt * 4) & 31);
#pragma unroll
                for (int r = 0; r < 4; ++r) dm4[r] = ((nib >> r) & 1u) ? a.drop.scale : 0.f;
            } else {
                drop_pair_x<DROP>(a.drop, hx + (uint32_t)(8 * kt) * ICKA_HASH_C0, dm4[0], dm4[1]);
                drop_pair_x<DROP>(a.drop, hx + (uint32_t)(8 * kt + 1) * ICKA_HASH_C0, dm4[2], dm4[3]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(s[qi][kt][r] - mx);
                psum += pv;
                s[qi][kt][r] = pv * dm4[r];
            }
        }
        if constexpr (KB) {
            if (q < a.Sq) {
                const int wpl = keep_wpl(a.Skv);
                uint32_t* kp = a.keepbits + (((int64_t)bh * a.Sq + q) * 4 + g) * wpl;
#pragma unroll
                for (int w = 0; w < (KT * 4 + 31) / 32; ++w)
                    if (w < wpl) kp[w] = kbw[w];
            }
        }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        inv[qi] = (FP8 ? 1.f / FP8_P_SCALE : 1.f) / psum;
        if (g == 0 && a.lse && q < a.Sq) a.lse[(int64_t)bh * a.Sq + q] = mx + logf(psum);
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            if constexpr (FP8) {
                const f32x4 lo = s[qi][2 * ks] * FP8_P_SCALE, hi = s[qi][2 * ks + 1] * FP8_P_SCALE;
                pf8[qi][ks] = fp8x8(lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]);
            } else {
                pf[qi][ks] = pack8(s[qi][2 * ks], s[qi][2 * ks + 1]);
            }
        }
    }
    f32x4 acc[QT][4];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[qi][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
            const bf16x8 vt = frag_tr(sV, dt, ks, lane);
            if constexpr (FP8) {
                const long v8 = fp8x8(vt);
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) acc[qi][dt] = mfma16_fp8(v8, pf8[qi][ks], acc[qi][dt]);
            } else {
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) acc[qi][dt] = mfma16(vt, pf[qi][ks], acc[qi][dt]);
            }
        }
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        const int q = q0 + 16 * (wq + qi) + i15;
        if (q < a.Sq) {
            bf16_t* orow = a.Ow + ((int64_t)b * a.Sq + q) * a.ldo + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<u32x2*>(orow + 16 * dt + 4 * g) =
                    pack4(acc[qi][dt][0] * inv[qi], acc[qi][dt][1] * inv[qi], acc[qi][dt][2] * inv[qi],
                          acc[qi][dt][3] * inv[qi]);
            if (a.Ow16) {   // (the context is a convex combination of value rows: no fp16 overflow beyond V's own range)
                _Float16* hrow = a.Ow16 + ((int64_t)b * a.Sq + q) * a.ldo + head * HD;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<f16x4*>(hrow + 16 * dt + 4 * g) =
                        f16x4{(_Float16)(acc[qi][dt][0] * inv[qi]), (_Float16)(acc[qi][dt][1] * inv[qi]),
                              (_Float16)(acc[qi][dt][2] * inv[qi]), (_Float16)(acc[qi][dt][3] * inv[qi])};
            }
        }
    }
}

}  // namespace
