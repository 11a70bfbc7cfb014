// The reference's parameter update after backward (My_cross_attention.py:831-844): clip_grad_norm_(model.parameters(), 1.0),
// AdamW.step() with the two weight-decay groups of :743-751, scheduler.step(), zero_grad.  Outside the fwd+bwd metric
//  but reported beside it; as ~220 per-tensor launches of the stock optimizer it costs 2.9 ms on a
// 4.5 ms step.  Parameters, gradients and optimizer state live in flat buffers of ONE layout (ParamArena), so the whole
// update is three launches over chunk tables:
//   icka_optim_sqnorm   per-chunk sums of squares of the gradients                      (reads 4 B per gradient)
//   icka_optim_clip     fixed-order sum of the partials -> total norm and the clip coefficient, on the device (no host sync)
//   icka_optim_adamw    p, m, v <- AdamW(p, g * coef, m, v) for one weight-decay group, and in the same pass the 16-bit
//                       weight shadows the next forward's GEMMs read (bf16, and fp16 in the "mixed16" mode): the separate
//                       per-forward re-cast of the arena (98 us at bert-base) has nothing left to do
// Arithmetic = torch.optim.AdamW (decoupled decay first, bias-corrected moments, eps added to sqrt(v_hat)); all f32.
#include "common.h"

namespace {

constexpr int OPT_CHUNK = 8192;

// table[2 b] = first element, table[2 b + 1] = count (multiple of 8, <= OPT_CHUNK)
__global__ __launch_bounds__(256) void optim_sqnorm_kernel(const float* __restrict__ g, const int64_t* __restrict__ table,
                                                           float* __restrict__ partials) {
    __shared__ float red[4];
    const int64_t lo = table[2 * blockIdx.x], len = table[2 * blockIdx.x + 1];
    float s = 0.f;
    for (int64_t c = threadIdx.x * 4; c < len; c += 256 * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(g + lo + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = total gradient norm, out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (clip_grad_norm_'s formula);
// one block, fixed summation order: bitwise repeatable
__global__ __launch_bounds__(1024) void optim_clip_kernel(const float* __restrict__ partials, int n, float max_norm,
                                                          float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) s += (double)partials[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += red[i];
        const float norm = (float)sqrt(t);
        out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);
        out[1] = max_norm > 0.f ? (c < 1.f ? c : 1.f) : 1.f;
    }
}

struct AdamArgs {
    float* p; const float* g; float* m; float* v;
    bf16_t* shadow; _Float16* shadow16;      // may be NULL
    const int64_t* table;
    const float* coef;                       // device scalar (out[1] of the clip kernel) or NULL
    float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt;   // bc1 = 1 - beta1^t, bc2_sqrt = sqrt(1 - beta2^t)
};

__global__ __launch_bounds__(256) void optim_adamw_kernel(const AdamArgs a) {
    const int64_t lo = a.table[2 * blockIdx.x], len = a.table[2 * blockIdx.x + 1];
    const float coef = a.coef ? a.coef[0] : 1.f;
    const float step = a.lr / a.bc1, decay = 1.f - a.lr * a.wd;
    for (int64_t c = threadIdx.x * 8; c < len; c += 256 * 8) {
        const int64_t i = lo + c;
        float p[8], g[8], m[8], v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 pv = *reinterpret_cast<const f32x4*>(a.p + i + 4 * h), gv = *reinterpret_cast<const f32x4*>(a.g + i + 4 * h);
            const f32x4 mv = *reinterpret_cast<const f32x4*>(a.m + i + 4 * h), vv = *reinterpret_cast<const f32x4*>(a.v + i + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { p[4 * h + e] = pv[e]; g[4 * h + e] = gv[e] * coef; m[4 * h + e] = mv[e]; v[4 * h + e] = vv[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            p[e] *= decay;                                               // decoupled weight decay (torch.optim.AdamW)
            m[e] = a.beta1 * m[e] + (1.f - a.beta1) * g[e];              // exp_avg.lerp_(grad, 1 - beta1)
            v[e] = a.beta2 * v[e] + (1.f - a.beta2) * g[e] * g[e];
            const float denom = sqrtf(v[e]) / a.bc2_sqrt + a.eps;
            p[e] -= step * (m[e] / denom);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<f32x4*>(a.p + i + 4 * h) = f32x4{p[4 * h], p[4 * h + 1], p[4 * h + 2], p[4 * h + 3]};
            *reinterpret_cast<f32x4*>(a.m + i + 4 * h) = f32x4{m[4 * h], m[4 * h + 1], m[4 * h + 2], m[4 * h + 3]};
            *reinterpret_cast<f32x4*>(a.v + i + 4 * h) = f32x4{v[4 * h], v[4 * h + 1], v[4 * h + 2], v[4 * h + 3]};
        }
        if (a.shadow) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(p[e]);
            *reinterpret_cast<u32x4*>(a.shadow + i) = as_u32x4(o);
        }
        if (a.shadow16) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (_Float16)fminf(fmaxf(p[e], -65504.f), 65504.f);
            *reinterpret_cast<f16x8*>(a.shadow16 + i) = o;
        }
    }
}

}  // namespace

extern "C" int64_t icka_optim_chunk_elems(void) { return OPT_CHUNK; }

extern "C" int icka_optim_sqnorm(const float* grads, const int64_t* table_dev, int32_t n_chunks, float* partials, void* stream) {
    if (!grads || !table_dev || !partials) return ICKA_E_ARG;
    if (n_chunks <= 0) return ICKA_E_SHAPE;
    if (reinterpret_cast<uintptr_t>(grads) & 15) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(optim_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, grads, table_dev, partials);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_optim_clip(const float* partials, int32_t n, float max_norm, float* out2, void* stream) {
    if (!partials || !out2) return ICKA_E_ARG;
    if (n <= 0) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(optim_clip_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, partials, n, max_norm, out2);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_optim_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16,
                                void* shadow_f16, const int64_t* table_dev, int32_t n_chunks, const float* clip_coef, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !table_dev) return ICKA_E_ARG;
    if (n_chunks <= 0 || step <= 0) return ICKA_E_SHAPE;
    if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(exp_avg) |
         reinterpret_cast<uintptr_t>(exp_avg_sq) | reinterpret_cast<uintptr_t>(shadow_bf16) | reinterpret_cast<uintptr_t>(shadow_f16)) & 15)
        return ICKA_E_ALIGN;
    AdamArgs a;
    a.p = params; a.g = grads; a.m = exp_avg; a.v = exp_avg_sq; a.shadow = (bf16_t*)shadow_bf16; a.shadow16 = (_Float16*)shadow_f16;
    a.table = table_dev; a.coef = clip_coef; a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.wd = weight_decay;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(optim_adamw_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}
