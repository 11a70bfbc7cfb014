// Shared device helpers for the ICKA MI355X (gfx950 / CDNA4) kernels.
// wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16; LDS images and the counter-hash dropout RNG live here so that
// forward and backward kernels regenerate identical masks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/icka_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define ICKA_WAVE 64
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

#define ICKA_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return (int)e__;               \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// bf16 <-> f32
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32, RNE, NaN-preserving

// NOTE: register arrays use clang ext_vector types (u32x4 ...), never HIP's uint4/float4 structs: arrays of the
// struct types are not always scalarised and end up in scratch / promoted to LDS.
__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ u32x4 as_u32x4(bf16x8 v) { return __builtin_bit_cast(u32x4, v); }
__device__ __forceinline__ bf16x4 as_bf16x4(u32x2 v) { return __builtin_bit_cast(bf16x4, v); }
__device__ __forceinline__ u32x2 as_u32x2(bf16x4 v) { return __builtin_bit_cast(u32x2, v); }

__device__ __forceinline__ u32x2 pack4(float a, float b, float c, float d) {
    bf16x4 v = {f2bf(a), f2bf(b), f2bf(c), f2bf(d)};
    return as_u32x2(v);
}

// ---------------------------------------------------------------------------------------------------------------
// wave-level reductions (64 lanes; call from wave-uniform code: every lane takes part and gets the result).
// Rows of 16 lanes reduce on the DPP path (row_ror: 4 VALU ops, every lane of a row holds the row's result), the four rows
// combine through lane reads.  (__shfl_xor compiles to ds_bpermute_b32 here: six dependent LDS-crossbar round trips of
// ~100 cycles each sat between the load and the store phase of every LayerNorm row.)
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_f32(float v, int lane) {   // value of a (compile-time) lane as a scalar operand
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_row<0x128>(v);   // row_ror:8
    v += dpp_row<0x124>(v);   // row_ror:4
    v += dpp_row<0x122>(v);   // row_ror:2
    v += dpp_row<0x121>(v);   // row_ror:1
    return (lane_f32(v, 0) + lane_f32(v, 16)) + (lane_f32(v, 32) + lane_f32(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_row<0x128>(v));
    v = fmaxf(v, dpp_row<0x124>(v));
    v = fmaxf(v, dpp_row<0x122>(v));
    v = fmaxf(v, dpp_row<0x121>(v));
    return fmaxf(fmaxf(lane_f32(v, 0), lane_f32(v, 16)), fmaxf(lane_f32(v, 32), lane_f32(v, 48)));
}

// ---------------------------------------------------------------------------------------------------------------
// Counter-hash dropout RNG.  keep(idx) is a pure function of (seed, element index) so that the backward kernels
// regenerate the forward mask in whatever register layout they hold the element (rows on lanes or in-lane).
struct DropCfg {
    uint32_t s0, s1;   // 64-bit seed halves
    uint32_t thr;      // drop if hash < thr ; thr = p * 2^32
    float scale;       // 1 / (1 - p)
    const uint32_t* nonce;  // optional device words XORed into the seed at kernel start: a captured hipGraph replays
                            // the same seed VALUES, the nonce (bumped by a node of the graph) makes every replay draw
                            // fresh masks while forward and backward of one step still agree
};
extern const uint32_t* g_icka_nonce;   // set by icka_set_dropout_nonce (elementwise.hip)
// CUs the caller has reserved for OTHER kernels that may run beside ours (icka_lstm_set_reserved_cus: dp.GradReducer reserves
// the CUs RCCL's workgroups may hold): every launch whose blocks wait for each other (the persistent BiLSTM, icka_gemm_ln)
// is used only when its grid fits into the device's CUs minus this reserve.  Defined in elementwise.hip.
extern int g_icka_reserved_cus;
__host__ __device__ __forceinline__ DropCfg make_drop(float p, uint64_t seed) {
    DropCfg d;
    d.s0 = (uint32_t)seed;
    d.s1 = (uint32_t)(seed >> 32);
#ifndef __HIP_DEVICE_COMPILE__
    d.nonce = g_icka_nonce;
#else
    d.nonce = nullptr;
#endif
    if (p <= 0.f) { d.thr = 0u; d.scale = 1.f; }
    else {
        double t = (double)p * 4294967296.0;
        d.thr = t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
        d.scale = 1.f / (1.f - p);
    }
    return d;
}
// hash(idx) = tail(idx * C0 + s0): kernels whose element index is base + compile-time offset keep base*C0 + s0 in a
// register and add offset*C0 (a constant), which removes one quarter-rate 32-bit multiply per element.
constexpr uint32_t ICKA_HASH_C0 = 0x9E3779B1u;
__host__ __device__ __forceinline__ uint32_t icka_hash_tail(uint32_t x, uint32_t s1) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x = x * 0x846ca68bu + s1;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t icka_hash(uint32_t s0, uint32_t s1, uint32_t idx) {
    return icka_hash_tail(idx * ICKA_HASH_C0 + s0, s1);
}
// fold the device nonce into the seed (call once at kernel entry)
__device__ __forceinline__ DropCfg drop_resolve(DropCfg d) {
    if (d.nonce && d.thr) { d.s0 ^= d.nonce[0]; d.s1 ^= d.nonce[1]; }
    return d;
}
// multiplier applied to a kept element: scale if kept, 0 if dropped
__device__ __forceinline__ float drop_mul(const DropCfg& d, uint32_t idx) {
    return (d.thr == 0u || icka_hash(d.s0, d.s1, idx) >= d.thr) ? d.scale : 0.f;
}

// same, with the "dropout off" case resolved at compile time (branch-free inner loops)
template <bool DROP>
__device__ __forceinline__ float drop_mul_t(const DropCfg& d, uint32_t idx) {
    if (!DROP) return 1.f;
    return icka_hash(d.s0, d.s1, idx) >= d.thr ? d.scale : 0.f;
}
// x = idx * ICKA_HASH_C0 + s0 already formed by the caller
template <bool DROP>
__device__ __forceinline__ float drop_mul_x(const DropCfg& d, uint32_t x) {
    if (!DROP) return 1.f;
    return icka_hash_tail(x, d.s1) >= d.thr ? d.scale : 0.f;
}

// Attention-probability dropout draws TWO decisions from one 32-bit hash (the whole-head attention kernels are VALU-bound
// and a third of their backward was hashing): element (row, key) of a probability matrix whose rows start at flat index
// idx_row = row * Skv uses hash(idx_row + key / 2) -- the low 16 bits for even keys, the high 16 bits for odd keys -- against
// thr >> 16 (p = 0.1: 6553 / 65536).  Every site that touches an attention mask (forward, both backward layouts, the tiled
// kernels, the fp32-mode softmax, icka_attn_dropout_mask) goes through these two functions.
__device__ __forceinline__ void drop_pair(const DropCfg& d, uint32_t pidx, float& m_even, float& m_odd) {
    const uint32_t h = icka_hash(d.s0, d.s1, pidx), t = d.thr >> 16;
    m_even = (h & 0xffffu) >= t ? d.scale : 0.f;
    m_odd = (h >> 16) >= t ? d.scale : 0.f;
}
// x = pidx * ICKA_HASH_C0 + s0 already formed by the caller
template <bool DROP>
__device__ __forceinline__ void drop_pair_x(const DropCfg& d, uint32_t x, float& m_even, float& m_odd) {
    if (!DROP) { m_even = m_odd = 1.f; return; }
    const uint32_t h = icka_hash_tail(x, d.s1), t = d.thr >> 16;
    m_even = (h & 0xffffu) >= t ? d.scale : 0.f;
    m_odd = (h >> 16) >= t ? d.scale : 0.f;
}
__device__ __forceinline__ float drop_mul_key(const DropCfg& d, uint32_t idx_row, uint32_t key) {
    float e, o;
    drop_pair(d, idx_row + (key >> 1), e, o);
    return (key & 1u) ? o : e;
}
// the four in-lane keys kb .. kb + 3 (kb % 4 == 0) of an MFMA accumulator
__device__ __forceinline__ void drop_mul_key4(const DropCfg& d, uint32_t idx_row, uint32_t kb, float (&m)[4]) {
    if (d.thr == 0u) { m[0] = m[1] = m[2] = m[3] = d.scale; return; }
    drop_pair(d, idx_row + (kb >> 1), m[0], m[1]);
    drop_pair(d, idx_row + (kb >> 1) + 1u, m[2], m[3]);
}

// ---------------------------------------------------------------------------------------------------------------
// erf-GELU (Cross_Modal_Interaction_Module.py:31-37) and its derivative.
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 grid of the outputs): one v_rcp, one v_exp
// and five FMAs instead of libm's erff -- the GEMM epilogues evaluate it 16K times per 128x128 tile.
// Returns erf(x/sqrt2) and, through `e`, exp(-x*x/2) (shared with the Gaussian term of the derivative).
__device__ __forceinline__ float erf_gauss(float x, float& e) {
    const float u = fabsf(x) * 0.70710678118654752f;
    const float t = __frcp_rn(1.f + 0.3275911f * u);
    e = __expf(-u * u);
    float p = 1.061405429f;
    p = p * t - 1.453152027f;
    p = p * t + 1.421413741f;
    p = p * t - 0.284496736f;
    p = p * t + 0.254829592f;
    return copysignf(1.f - p * t * e, x);
}
#ifndef ICKA_GELU_EXACT
// Default: polynomial forms for the GEMM epilogues (build with -DICKA_GELU_EXACT for the erf form below) (no v_rcp / v_exp: both are quarter-rate): Phi(x) and gelu'(x) = Phi(x) +
// x*phi(x) are 0.5 + odd functions; minimax fits 0.5 + xc*P(xc^2) on |x| <= 4 (8 coefficients), xc = clamp(x, -4, 4).
// |error| <= 5.4e-5 (Phi) / 2.8e-4 (gelu'), against a bf16 output grid of 3.9e-3 relative; the compiler packs the Horner
// chains of neighbouring elements into v_pk_fma_f32.  Same-box A/B on the c2 step: +1.5 % (the erf form's rcp + exp cost
// ~7 us of VALU time per 12.6 M-element FFN GEMM).
__device__ __forceinline__ float odd_poly8(float x, const float (&c)[8]) {
    const float xc = fminf(fmaxf(x, -4.f), 4.f);
    const float t = xc * xc;
    float p = c[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) p = fmaf(p, t, c[k]);
    return fmaf(xc, p, 0.5f);
}
__device__ __forceinline__ float gelu_f(float x) {
    const float c[8] = {3.988475103e-01f, -6.617537857e-02f, 9.664874089e-03f, -1.048204393e-03f,
                        8.066739589e-05f, -4.100866561e-06f, 1.217111155e-07f, -1.580786406e-09f};
    return x < -4.f ? 0.f : x * odd_poly8(x, c);   // gelu(-4) = -1.3e-4; below, the clamped Phi would leave |x| * 8e-5
}
__device__ __forceinline__ float dgelu_f(float x) {
    const float c[8] = {7.967216565e-01f, -2.620298586e-01f, 5.591483287e-02f, -7.687443586e-03f,
                        6.876457098e-04f, -3.845959300e-05f, 1.213807069e-06f, -1.641980264e-08f};
    return odd_poly8(x, c);
}
#else
__device__ __forceinline__ float gelu_f(float x) {
    float e;
    return 0.5f * x * (1.f + erf_gauss(x, e));
}
__device__ __forceinline__ float dgelu_f(float x) {
    float e;
    const float er = erf_gauss(x, e);
    return 0.5f * (1.f + er) + x * 0.3989422804014327f * e;
}
#endif
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------------------------
// MFMA: D[i][j] += sum_k a[i][k] * b[k][j], 16x16x32 bf16.
//   lane l holds a[i = l&15][k = 8*(l>>4) + e], b[k = 8*(l>>4) + e][j = l&15], e = 0..7,
//   and D[i = 4*(l>>4) + r][j = l&15], r = 0..3.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// the same fragment registers holding IEEE fp16 operands (v_mfma_f32_16x16x32_f16, same rate and layout)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
template <bool F16>
__device__ __forceinline__ f32x4 mfma16t(bf16x8 a, bf16x8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return mfma16(a, b, c);
}

// LDS reads
__device__ __forceinline__ bf16x8 lds_read_b128(const char* base, uint32_t off) {
    return *reinterpret_cast<const bf16x8*>(base + off);
}
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-col block of 16-bit elements is delivered column-major:
// lane 4q+p supplies the address of row q, cols 4p..4p+3; lane i receives column i, rows 0..3.
__device__ __forceinline__ bf16x4 lds_read_tr(const char* base, uint32_t off) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, base + off));
}
__device__ __forceinline__ bf16x8 join8(bf16x4 lo, bf16x4 hi) {
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
