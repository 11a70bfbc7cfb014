// Per-sample gates of the two MNER heads (gfx950), all HBM-bound element-wise / reduction kernels:
//   * Cross_Modal form (Cross_Modal_Interaction_Module.py:1029-1036): g_b = sigmoid(logit_b);
//         out[b,s,:] = g_b * tok[b,s,:] + (1 - g_b) * cross[b,s,:]
//   * gate_cl form (my_bert/gate_cl_modeling.py:1364-1373): crs[b,:] = Linear_{2HS->2}(cat(seq,cross)[b].view(-1));
//         p_b = softmax(crs[b])[-1];  cross'[b] = p_b * cross[b]
// Token-major bf16 matrices [B*S, H] with row strides; 16-byte chunks per lane; per-sample sums by block reduction +
// one f32 atomic per block (the accumulators are zeroed by the caller).
#include <math.h>

#include "common.h"

namespace {

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
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    __syncthreads();
    return t;
}
inline bool ok16(const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; }

constexpr int SPLIT = 8;  // blocks per sample

// mode 0: g = sigmoid(gate[b]) ; out = g*a + (1-g)*c         (blend; c may be NULL -> out = g*a)
// mode 1: g = softmax(gate[b, 0..1])[1] ; out = g*a
__device__ __forceinline__ float sample_gate(const float* gate, int b, int mode) {
    if (mode == 0) return 1.f / (1.f + __expf(-gate[b]));
    const float x0 = gate[2 * b], x1 = gate[2 * b + 1];
    return 1.f / (1.f + __expf(x0 - x1));
}

__global__ __launch_bounds__(256) void gate_fwd_kernel(const bf16_t* __restrict__ a, int64_t lda,
                                                       const bf16_t* __restrict__ c, int64_t ldc,
                                                       const float* __restrict__ gate, int mode,
                                                       bf16_t* __restrict__ out, int64_t ldo, int S, int H) {
    const int b = blockIdx.y;
    const float g = sample_gate(gate, b, mode);
    const int cpr = H >> 3, total = S * cpr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int s = i / cpr, ch = (i - s * cpr) * 8;
        const int64_t row = (int64_t)b * S + s;
        float x[8], y[8];
        ld8(a + row * lda + ch, x);
        if (c) {
            ld8(c + row * ldc + ch, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = g * x[e] + (1.f - g) * y[e];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] *= g;
        }
        st8(out + row * ldo + ch, x);
    }
}

// "mixed16" form of mode 1 (the relevance-scaled cross stream of the gate_cl head, gate_cl_modeling.py:1369-1373): the fp16
// twin of the stream in, g * a out in BOTH 16-bit types -- fp16 (operand of the gate GEMM and the classifier) and bf16 (what
// the backward reads) -- from the f32 product, so the fp16 copy carries no bf16 rounding
__global__ __launch_bounds__(256) void gate_fwd_h_kernel(const _Float16* __restrict__ a, int64_t lda,
                                                         const float* __restrict__ gate, int mode,
                                                         bf16_t* __restrict__ out, _Float16* __restrict__ out16, int64_t ldo,
                                                         int S, int H) {
    const int b = blockIdx.y;
    const float g = sample_gate(gate, b, mode);
    const int cpr = H >> 3, total = S * cpr;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int s = i / cpr, ch = (i - s * cpr) * 8;
        const int64_t row = (int64_t)b * S + s;
        const f16x8 xv = *reinterpret_cast<const f16x8*>(a + row * lda + ch);
        float x[8];
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            x[e] = g * (float)xv[e];
            h[e] = (_Float16)fminf(fmaxf(x[e], -65504.f), 65504.f);
        }
        st8(out + row * ldo + ch, x);
        *reinterpret_cast<f16x8*>(out16 + row * ldo + ch) = h;
    }
}

// backward of the above: da = g*dout ; dc = (1-g)*dout (mode 0 with c) ;
// dgate: mode 0: dgate[b] += g(1-g) * sum dout*(a-c) ; mode 1: dgate[b,1] += t, dgate[b,0] -= t, t = g(1-g)*sum dout*a
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* __restrict__ dout, int64_t lddo,
                                                       const bf16_t* __restrict__ a, int64_t lda,
                                                       const bf16_t* __restrict__ c, int64_t ldc,
                                                       const float* __restrict__ gate, int mode,
                                                       bf16_t* __restrict__ da, int64_t ldda,
                                                       bf16_t* __restrict__ dc, int64_t lddc,
                                                       float* __restrict__ dgate, int S, int H) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float g = sample_gate(gate, b, mode);
    const int cpr = H >> 3, total = S * cpr;
    float acc = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int s = i / cpr, ch = (i - s * cpr) * 8;
        const int64_t row = (int64_t)b * S + s;
        float d[8], x[8], y[8], o[8];
        ld8(dout + row * lddo + ch, d);
        ld8(a + row * lda + ch, x);
        if (c) {
            ld8(c + row * ldc + ch, y);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += d[e] * (x[e] - y[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += d[e] * x[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = g * d[e];
        st8(da + row * ldda + ch, o);
        if (c && dc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (1.f - g) * d[e];
            st8(dc + row * lddc + ch, o);
        }
    }
    const float t = block_sum(acc, red) * g * (1.f - g);
    if (threadIdx.x == 0) {
        if (mode == 0) atomicAdd(dgate + b, t);
        else { atomicAdd(dgate + 2 * b + 1, t); atomicAdd(dgate + 2 * b, -t); }
    }
}

// crs[b,j] += sum_{s,k} seq[b,s,k]*W[j, s*2H + k] + cross[b,s,k]*W[j, s*2H + H + k]   (+ bias_j once)
__global__ __launch_bounds__(256) void crs_fwd_kernel(const bf16_t* __restrict__ seq, int64_t lds,
                                                      const bf16_t* __restrict__ cross, int64_t ldc,
                                                      const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ crs, int S, int H) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const int cpr = H >> 3, total = S * cpr;
    const int64_t wrow = (int64_t)S * 2 * H;
    float a0 = 0.f, a1 = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int s = i / cpr, ch = (i - s * cpr) * 8;
        const int64_t row = (int64_t)b * S + s;
        float x[8], y[8], w[8];
        ld8(seq + row * lds + ch, x);
        ld8(cross + row * ldc + ch, y);
        const int64_t wo = (int64_t)s * 2 * H + ch;
        ld8(W + wo, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) a0 += x[e] * w[e];
        ld8(W + wo + H, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) a0 += y[e] * w[e];
        ld8(W + wrow + wo, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) a1 += x[e] * w[e];
        ld8(W + wrow + wo + H, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) a1 += y[e] * w[e];
    }
    a0 = block_sum(a0, red);
    a1 = block_sum(a1, red);
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0 && bias) { a0 += bias[0]; a1 += bias[1]; }
        atomicAdd(crs + 2 * b, a0);
        atomicAdd(crs + 2 * b + 1, a1);
    }
}

// input gradients of the crs Linear: dseq[b,s,k] = sum_j dcrs[b,j] W[j, s*2H+k] ; dcross likewise with +H
__global__ __launch_bounds__(256) void crs_dgrad_kernel(const float* __restrict__ dcrs, const bf16_t* __restrict__ W,
                                                        bf16_t* __restrict__ dseq, bf16_t* __restrict__ dcross, int B,
                                                        int S, int H) {
    const int cpr = H >> 3;
    const int64_t total = (int64_t)B * S * cpr, wrow = (int64_t)S * 2 * H;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr) * 8;
        const int b = (int)(row / S), s = (int)(row - (int64_t)b * S);
        const float d0 = dcrs[2 * b], d1 = dcrs[2 * b + 1];
        const int64_t wo = (int64_t)s * 2 * H + ch;
        float w0[8], w1[8], o[8];
        ld8(W + wo, w0); ld8(W + wrow + wo, w1);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = d0 * w0[e] + d1 * w1[e];
        st8(dseq + row * H + ch, o);
        ld8(W + wo + H, w0); ld8(W + wrow + wo + H, w1);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = d0 * w0[e] + d1 * w1[e];
        st8(dcross + row * H + ch, o);
    }
}

// weight gradient: dW[j, s*2H + k] (+)= sum_b dcrs[b,j] seq[b,s,k] ; second half from cross.  One thread per 8 k.
__global__ __launch_bounds__(256) void crs_wgrad_kernel(const float* __restrict__ dcrs, const bf16_t* __restrict__ seq,
                                                        int64_t lds, const bf16_t* __restrict__ cross, int64_t ldc,
                                                        float* __restrict__ dW, float* __restrict__ dbias, int B, int S,
                                                        int H, int accumulate) {
    const int cpr = H >> 3;
    const int64_t total = (int64_t)S * cpr * 2, wrow = (int64_t)S * 2 * H;   // (s, half, chunk)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int s = (int)(i / (2 * cpr));
        const int r = (int)(i - (int64_t)s * 2 * cpr);
        const int half = r / cpr, ch = (r - half * cpr) * 8;
        const bf16_t* src = half ? cross : seq;
        const int64_t ld = half ? ldc : lds;
        float g0[8], g1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { g0[e] = 0.f; g1[e] = 0.f; }
        for (int b = 0; b < B; ++b) {
            float x[8];
            ld8(src + ((int64_t)b * S + s) * ld + ch, x);
            const float d0 = dcrs[2 * b], d1 = dcrs[2 * b + 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) { g0[e] += d0 * x[e]; g1[e] += d1 * x[e]; }
        }
        float* p0 = dW + (int64_t)s * 2 * H + half * H + ch;
        float* p1 = p0 + wrow;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p0[e] = accumulate ? p0[e] + g0[e] : g0[e];
            p1[e] = accumulate ? p1[e] + g1[e] : g1[e];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 2 && dbias) {
        float t = 0.f;
        for (int b = 0; b < B; ++b) t += dcrs[2 * b + threadIdx.x];
        dbias[threadIdx.x] = accumulate ? dbias[threadIdx.x] + t : t;
    }
}

}  // namespace

extern "C" int icka_sample_gate_fwd(const void* a, int64_t lda, const void* c, int64_t ldc, const float* gate,
                                    int32_t mode, void* out, int64_t ldo, int32_t B, int32_t S, int32_t H,
                                    void* stream) {
    if (!a || !gate || !out || (mode != 0 && mode != 1)) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8) return ICKA_E_SHAPE;
    if (!ok16(a, lda) || !ok16(out, ldo) || (c && !ok16(c, ldc))) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(SPLIT, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a, lda,
                       (const bf16_t*)c, ldc, gate, mode, (bf16_t*)out, ldo, S, H);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_sample_gate_fwd_h(const void* a16, int64_t lda, const float* gate, int32_t mode, void* out, void* out16,
                                      int64_t ldo, int32_t B, int32_t S, int32_t H, void* stream) {
    if (!a16 || !gate || !out || !out16) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8 != 0 || (mode != 0 && mode != 1)) return ICKA_E_SHAPE;
    if (!ok16(a16, lda) || !ok16(out, ldo) || !ok16(out16, ldo)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(gate_fwd_h_kernel, dim3(SPLIT, B), dim3(256), 0, (hipStream_t)stream, (const _Float16*)a16, lda, gate,
                       mode, (bf16_t*)out, (_Float16*)out16, ldo, S, H);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_sample_gate_bwd(const void* dout, int64_t lddo, const void* a, int64_t lda, const void* c,
                                    int64_t ldc, const float* gate, int32_t mode, void* da, int64_t ldda, void* dc,
                                    int64_t lddc, float* dgate, int32_t B, int32_t S, int32_t H, void* stream) {
    if (!dout || !a || !gate || !da || !dgate || (mode != 0 && mode != 1)) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8) return ICKA_E_SHAPE;
    if (!ok16(dout, lddo) || !ok16(a, lda) || !ok16(da, ldda) || (c && !ok16(c, ldc)) || (dc && !ok16(dc, lddc)))
        return ICKA_E_ALIGN;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(SPLIT, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout, lddo,
                       (const bf16_t*)a, lda, (const bf16_t*)c, ldc, gate, mode, (bf16_t*)da, ldda, (bf16_t*)dc, lddc,
                       dgate, S, H);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_crs_fwd(const void* seq, int64_t lds, const void* cross, int64_t ldc, const void* W,
                            const float* bias, float* crs, int32_t B, int32_t S, int32_t H, void* stream) {
    if (!seq || !cross || !W || !crs) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8) return ICKA_E_SHAPE;
    if (!ok16(seq, lds) || !ok16(cross, ldc) || !ok16(W, 8)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(crs_fwd_kernel, dim3(SPLIT, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)seq, lds,
                       (const bf16_t*)cross, ldc, (const bf16_t*)W, bias, crs, S, H);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_crs_bwd(const float* dcrs, const void* seq, int64_t lds, const void* cross, int64_t ldc,
                            const void* W, void* dseq, void* dcross, float* dW, float* dbias, int32_t B, int32_t S,
                            int32_t H, int32_t accumulate, void* stream) {
    if (!dcrs || !seq || !cross || !W || !dseq || !dcross || !dW) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8) return ICKA_E_SHAPE;
    if (!ok16(seq, lds) || !ok16(cross, ldc) || !ok16(W, 8) || !ok16(dseq, 8) || !ok16(dcross, 8)) return ICKA_E_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n1 = (int64_t)B * S * (H / 8);
    hipLaunchKernelGGL(crs_dgrad_kernel, dim3((unsigned)((n1 + 255) / 256 > 2048 ? 2048 : (n1 + 255) / 256)), dim3(256),
                       0, st, dcrs, (const bf16_t*)W, (bf16_t*)dseq, (bf16_t*)dcross, B, S, H);
    ICKA_CHECK_LAUNCH();
    const int64_t n2 = (int64_t)S * (H / 8) * 2;
    hipLaunchKernelGGL(crs_wgrad_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, dcrs,
                       (const bf16_t*)seq, lds, (const bf16_t*)cross, ldc, dW, dbias, B, S, H, accumulate);
    ICKA_CHECK_LAUNCH();
    return 0;
}


// ===================================================================================================================
// Classifier of the gated head (cl_modeling.py:1371: logits = classifier(cat(seq, Gate*cross))), C <= 16 labels.
// A 13-wide output is a poor fit for 128x128 MFMA tiles (the general GEMM path spent ~120 us per step on it: forward
// 16 us + split-K prescale, two K = 13 dgrad GEMMs of 22 us, two M = 13 wgrad GEMMs + prescales, a column-sum pair,
// and the gate backward pass); here it is two HBM-bound kernels:
//   forward : a wave per token row, lanes over 16-byte chunks of [seq | gated], W (bf16 [C, 2H]) in LDS
//   backward: a block per 32 tokens, a thread per 16-byte chunk of the 2H columns holding W[:, chunk] and dW[:, chunk]
//             in registers: dseq_c, and -- for the gated half -- the gate backward (du, dcross) computed in place of
//             dgated; dW / db leave as one slab per block, summed by a slab reduction (icka_gemm_grouped_ex).
namespace {

constexpr int CLS_TB = 16;      // tokens per backward block
constexpr int CLS_MAXC = 16;

__device__ __forceinline__ void cls_ld8(const bf16_t* p, float (&o)[8]) {
    const bf16x8 v = as_bf16x8(*reinterpret_cast<const u32x4*>(p));
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(v[e]);
}
__device__ __forceinline__ void cls_st8(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    *reinterpret_cast<u32x4*>(p) = as_u32x4(o);
}

// F16: seq, gated and W are IEEE fp16 (the "mixed16" head: the classifier reads the fp16 twins, 11 significand bits)
template <bool F16>
__global__ __launch_bounds__(256) void cls_head_fwd_kernel(const bf16_t* __restrict__ seq, const bf16_t* __restrict__ gated,
                                                           const bf16_t* __restrict__ W, const float* __restrict__ bias,
                                                           float* __restrict__ logits, int M, int H, int C) {
    extern __shared__ __attribute__((aligned(16))) char cls_smem[];
    __shared__ float s_red[4][CLS_MAXC][65];
    bf16_t* sW = reinterpret_cast<bf16_t*>(cls_smem);   // [C][2H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nch = (2 * H) >> 3, hch = H >> 3;
    for (int i = tid; i < C * nch; i += 256)
        *reinterpret_cast<u32x4*>(sW + (int64_t)i * 8) = *reinterpret_cast<const u32x4*>(W + (int64_t)i * 8);
    __syncthreads();
    const int nw = gridDim.x * 4;
    for (int row0 = (blockIdx.x * 4 + wave) * 2; row0 < M; row0 += 2 * nw) {
        // two token rows per pass; their (up to 4 per lane) 16-byte chunks are all requested before any use
        u32x4 xr[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = lane + 64 * i, row = row0 + r;
                xr[r][i] = u32x4{0u, 0u, 0u, 0u};
                if (j < nch && row < M)
                    xr[r][i] = *reinterpret_cast<const u32x4*>(j < hch ? seq + (int64_t)row * H + j * 8
                                                                       : gated + (int64_t)row * H + (j - hch) * 8);
            }
        float acc[2][CLS_MAXC];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < CLS_MAXC; ++c) acc[r][c] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = lane + 64 * i;
            if (j < nch) {
#pragma unroll
                for (int c = 0; c < CLS_MAXC; ++c) {
                    if (c < C) {
                        float w[8];
                        if constexpr (F16) {
                            const f16x8 wv = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(sW + (int64_t)c * 2 * H + j * 8));
#pragma unroll
                            for (int e = 0; e < 8; ++e) w[e] = (float)wv[e];
                        } else {
                            cls_ld8(sW + (int64_t)c * 2 * H + j * 8, w);
                        }
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            if constexpr (F16) {
                                const f16x8 xv = __builtin_bit_cast(f16x8, xr[r][i]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) acc[r][c] += (float)xv[e] * w[e];
                            } else {
                                const bf16x8 xv = as_bf16x8(xr[r][i]);
#pragma unroll
                                for (int e = 0; e < 8; ++e) acc[r][c] += bf2f(xv[e]) * w[e];
                            }
                        }
                    }
                }
            }
        }
        // cross-lane sums through LDS (a [16 classes][64 lanes] transpose per row): 16 writes + 16 reads + 2 shuffles
        // instead of 16 x 6 ds_bpermute shuffles
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int c = 0; c < CLS_MAXC; ++c) s_red[wave][c][lane] = acc[r][c];
            __builtin_amdgcn_wave_barrier();
            const int c = lane >> 2, qd = lane & 3;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) v += s_red[wave][c][16 * qd + i];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if (qd == 0 && c < C && row0 + r < M)
                logits[(int64_t)(row0 + r) * C + c] = v + (bias ? bias[c] : 0.f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <int CC>
__global__ __launch_bounds__(256) void cls_head_bwd_kernel(const bf16_t* __restrict__ dl, int64_t ldd,
                                                           const bf16_t* __restrict__ seq, const bf16_t* __restrict__ gated,
                                                           const bf16_t* __restrict__ gate, const bf16_t* __restrict__ cross,
                                                           const bf16_t* __restrict__ W, bf16_t* __restrict__ dseq,
                                                           bf16_t* __restrict__ du, bf16_t* __restrict__ dcross,
                                                           float* __restrict__ partials, int M, int H, int C) {
    __shared__ float s_dl[CLS_TB][CLS_MAXC];
    const int tid = threadIdx.x;
    const int nch = (2 * H) >> 3, hch = H >> 3;
    const int t0 = blockIdx.x * CLS_TB;
    const int nt = M - t0 < CLS_TB ? M - t0 : CLS_TB;
    for (int i = tid; i < CLS_TB * CLS_MAXC; i += 256) {
        const int t = i / CLS_MAXC, c = i % CLS_MAXC;
        s_dl[t][c] = (t < nt && c < C) ? bf2f(dl[(int64_t)(t0 + t) * ldd + c]) : 0.f;
    }
    __syncthreads();
    const int64_t slab = (int64_t)C * 2 * H + CLS_MAXC;
    float* myslab = partials + (int64_t)blockIdx.x * slab;
    if (tid < CLS_MAXC) {   // bias gradient of this block's tokens
        float s = 0.f;
        for (int t = 0; t < nt; ++t) s += s_dl[t][tid];
        myslab[(int64_t)C * 2 * H + tid] = s;
    }
    const int j = tid;
    if (j >= nch) return;
    const bool second = j >= hch;
    const int h = (second ? j - hch : j) * 8;
    float Wr[CC][8], dW[CC][8];
#pragma unroll
    for (int c = 0; c < CC; ++c) {
        if (c < C) cls_ld8(W + (int64_t)c * 2 * H + j * 8, Wr[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { dW[c][e] = 0.f; if (c >= C) Wr[c][e] = 0.f; }
    }
    for (int tb = 0; tb < nt; tb += 4) {
        // 4 tokens per pass: their x (and gate / cross) chunks are requested before the arithmetic of the first
        u32x4 xr[4], gr[4], cr4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t row = t0 + (tb + u < nt ? tb + u : nt - 1);
            xr[u] = *reinterpret_cast<const u32x4*>((second ? gated : seq) + row * H + h);
            if (second) {
                gr[u] = *reinterpret_cast<const u32x4*>(gate + row * H + h);
                cr4[u] = *reinterpret_cast<const u32x4*>(cross + row * H + h);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = tb + u;
            if (t >= nt) break;
            const int64_t row = t0 + t;
            const bf16x8 xv = as_bf16x8(xr[u]);
            float x[8], dx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { x[e] = bf2f(xv[e]); dx[e] = 0.f; }
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float d = s_dl[t][c];
#pragma unroll
                for (int e = 0; e < 8; ++e) { dx[e] += d * Wr[c][e]; dW[c][e] += d * x[e]; }
            }
            if (!second) {
                cls_st8(dseq + row * H + h, dx);
            } else {   // dx is d(gate*cross): gate backward in place (cl_modeling.py:1363-1367)
                const bf16x8 gv = as_bf16x8(gr[u]), cv = as_bf16x8(cr4[u]);
                float uu[8], dc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float g = bf2f(gv[e]);
                    uu[e] = dx[e] * bf2f(cv[e]) * g * (1.f - g);
                    dc[e] = dx[e] * g;
                }
                cls_st8(du + row * H + h, uu);
                cls_st8(dcross + row * H + h, dc);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CC; ++c)
        if (c < C) {
            float* p = myslab + (int64_t)c * 2 * H + j * 8;
            *reinterpret_cast<f32x4*>(p) = f32x4{dW[c][0], dW[c][1], dW[c][2], dW[c][3]};
            *reinterpret_cast<f32x4*>(p + 4) = f32x4{dW[c][4], dW[c][5], dW[c][6], dW[c][7]};
        }
}

}  // namespace

extern "C" int32_t icka_cls_head_bwd_slabs(int32_t M) { return (M + CLS_TB - 1) / CLS_TB; }
extern "C" int64_t icka_cls_head_slab_floats(int32_t H, int32_t C) { return (int64_t)C * 2 * H + CLS_MAXC; }

static int cls_head_fwd_impl(bool f16, const void* seq, const void* gated, const void* W, const float* bias, float* logits,
                             int32_t M, int32_t H, int32_t C, void* stream) {
    if (!seq || !gated || !W || !logits) return ICKA_E_ARG;
    if (M <= 0 || H <= 0 || H % 8 || C <= 0 || C > CLS_MAXC || (int64_t)C * 2 * H * 2 > 44 * 1024) return ICKA_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(seq) | reinterpret_cast<uintptr_t>(gated) | reinterpret_cast<uintptr_t>(W)) & 15)
        return ICKA_E_ALIGN;
    int grid = (M + 7) / 8;   // 4 waves x 2 rows per pass
    grid = grid > 512 ? 512 : grid;
    if (f16)
        hipLaunchKernelGGL(cls_head_fwd_kernel<true>, dim3(grid), dim3(256), (size_t)C * 2 * H * 2, (hipStream_t)stream,
                           (const bf16_t*)seq, (const bf16_t*)gated, (const bf16_t*)W, bias, logits, M, H, C);
    else
        hipLaunchKernelGGL(cls_head_fwd_kernel<false>, dim3(grid), dim3(256), (size_t)C * 2 * H * 2, (hipStream_t)stream,
                           (const bf16_t*)seq, (const bf16_t*)gated, (const bf16_t*)W, bias, logits, M, H, C);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_cls_head_fwd(const void* seq, const void* gated, const void* W, const float* bias, float* logits,
                                 int32_t M, int32_t H, int32_t C, void* stream) {
    return cls_head_fwd_impl(false, seq, gated, W, bias, logits, M, H, C, stream);
}
// "mixed16" form: seq, gated and W are fp16
extern "C" int icka_cls_head_fwd_h(const void* seq, const void* gated, const void* W, const float* bias, float* logits,
                                   int32_t M, int32_t H, int32_t C, void* stream) {
    return cls_head_fwd_impl(true, seq, gated, W, bias, logits, M, H, C, stream);
}

extern "C" int icka_cls_head_bwd(const void* dl, int64_t ldd, const void* seq, const void* gated, const void* gate,
                                 const void* cross, const void* W, void* dseq, void* du, void* dcross, float* partials,
                                 int32_t M, int32_t H, int32_t C, void* stream) {
    if (!dl || !seq || !gated || !gate || !cross || !W || !dseq || !du || !dcross || !partials) return ICKA_E_ARG;
    if (M <= 0 || H <= 0 || H % 8 || C <= 0 || C > CLS_MAXC || 2 * H / 8 > 256 || ldd < C) return ICKA_E_SHAPE;
    const uintptr_t al = reinterpret_cast<uintptr_t>(seq) | reinterpret_cast<uintptr_t>(gated) |
                         reinterpret_cast<uintptr_t>(gate) | reinterpret_cast<uintptr_t>(cross) |
                         reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(dseq) |
                         reinterpret_cast<uintptr_t>(du) | reinterpret_cast<uintptr_t>(dcross) |
                         reinterpret_cast<uintptr_t>(partials);
    if (al & 15) return ICKA_E_ALIGN;
    const int grid = icka_cls_head_bwd_slabs(M);
    if (C <= 13)   // the reference's label sets have 13 (cl_modeling comments) or 15 tags: 13 saves 48 registers
        hipLaunchKernelGGL((cls_head_bwd_kernel<13>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dl, ldd, (const bf16_t*)seq, (const bf16_t*)gated, (const bf16_t*)gate,
                           (const bf16_t*)cross, (const bf16_t*)W, (bf16_t*)dseq, (bf16_t*)du, (bf16_t*)dcross,
                           partials, M, H, C);
    else
        hipLaunchKernelGGL((cls_head_bwd_kernel<CLS_MAXC>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)dl, ldd, (const bf16_t*)seq, (const bf16_t*)gated, (const bf16_t*)gate,
                           (const bf16_t*)cross, (const bf16_t*)W, (bf16_t*)dseq, (bf16_t*)du, (bf16_t*)dcross,
                           partials, M, H, C);
    ICKA_CHECK_LAUNCH();
    return 0;
}
