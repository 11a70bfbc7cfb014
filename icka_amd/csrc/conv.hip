// Data-movement kernels of the frozen ResNet-152 image encoder (resnet/resnet.py:57-150,
// resnet/resnet_utils.py:13-53), forward only.  Every convolution is a GEMM of the GEMM kernels on NHWC bf16
// activations ([B*H*W, C] matrices, eval-mode BatchNorm folded into the weights / bias, ReLU and the residual add in
// the GEMM epilogue: ICKA_EPI_RELU / ICKA_EPI_ADD_RELU); this file only builds the GEMM operands:
//   stem_patches : [B,3,H,W] f32 NCHW image  -> 7x7 / stride 2 / pad 3 patches, bf16 [B*Ho*Wo, 192] (147 taps + zero pad)
//   im2col3x3    : NHWC bf16 -> 3x3 / pad 1 patches, stride 1 or 2, [B*Ho*Wo, 9*C]
//   subsample    : NHWC rows at (s*y, s*x) for the strided 1x1 downsample convolutions
//   maxpool3x3s2 : nn.MaxPool2d(3, 2, 1)
//   features_out : last NHWC map -> att f32 [B,C,7,7] (F.adaptive_avg_pool2d(x,[7,7]) of a 7x7 map = identity),
//                  fc = mean over the 49 positions, and the region-token matrix bf16 [B*49, C] the MNER trunk reads.
// Rows past the last valid row of a padded (multiple-of-128) GEMM operand are written as zeros.
#include "common.h"

namespace {

__global__ void stem_patches_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, int B, int H, int W, int Ho,
                                    int Wo, int64_t rows_padded) {
    // one thread per (row, 8-element chunk of the 192-wide patch): k = (ky*7 + kx)*3 + c
    const int64_t total = rows_padded * 24;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / 24;
        const int ch = (int)(i - row * 24);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f2bf(0.f);
        if (row < (int64_t)B * Ho * Wo) {
            const int b = (int)(row / (Ho * Wo)), r = (int)(row - (int64_t)b * Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = ch * 8 + e;
                if (k < 147) {
                    const int tap = k / 3, c = k - tap * 3, ky = tap / 7, kx = tap - ky * 7;
                    const int y = oy * 2 - 3 + ky, x = ox * 2 - 3 + kx;
                    if (y >= 0 && y < H && x >= 0 && x < W) v[e] = f2bf(img[(((int64_t)b * 3 + c) * H + y) * W + x]);
                }
            }
        }
        *reinterpret_cast<u32x4*>(out + row * 192 + ch * 8) = as_u32x4(v);
    }
}

__global__ void im2col3x3_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ out, int B, int H, int W, int C,
                                 int stride, int Ho, int Wo, int64_t rows_padded) {
    const int cpr = C >> 3;                       // 16-byte chunks per tap
    const int64_t total = rows_padded * 9 * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (9 * cpr);
        const int rem = (int)(i - row * 9 * cpr), tap = rem / cpr, ch = rem - tap * cpr;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < (int64_t)B * Ho * Wo) {
            const int b = (int)(row / (Ho * Wo)), r = (int)(row - (int64_t)b * Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
            const int y = oy * stride - 1 + tap / 3, x = ox * stride - 1 + tap % 3;
            if (y >= 0 && y < H && x >= 0 && x < W)
                v = *reinterpret_cast<const u32x4*>(src + (((int64_t)b * H + y) * W + x) * C + ch * 8);
        }
        *reinterpret_cast<u32x4*>(out + row * 9 * C + (int64_t)tap * C + ch * 8) = v;
    }
}

__global__ void subsample_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ out, int B, int H, int W, int C,
                                 int stride, int Ho, int Wo, int64_t rows_padded) {
    const int cpr = C >> 3;
    const int64_t total = rows_padded * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row < (int64_t)B * Ho * Wo) {
            const int b = (int)(row / (Ho * Wo)), r = (int)(row - (int64_t)b * Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
            v = *reinterpret_cast<const u32x4*>(src + (((int64_t)b * H + oy * stride) * W + ox * stride) * C + ch * 8);
        }
        *reinterpret_cast<u32x4*>(out + row * C + ch * 8) = v;
    }
}

__global__ void maxpool3x3s2_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ out, int B, int H, int W, int C,
                                    int Ho, int Wo, int64_t rows_padded) {
    const int cpr = C >> 3;
    const int64_t total = rows_padded * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / cpr;
        const int ch = (int)(i - row * cpr);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(0.f);
        if (row < (int64_t)B * Ho * Wo) {
            const int b = (int)(row / (Ho * Wo)), r = (int)(row - (int64_t)b * Ho * Wo), oy = r / Wo, ox = r - oy * Wo;
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = oy * 2 - 1 + ky, x = ox * 2 - 1 + kx;
                    if (y < 0 || y >= H || x < 0 || x >= W) continue;   // padding never wins (nn.MaxPool2d pads with -inf)
                    const bf16x8 v = as_bf16x8(*reinterpret_cast<const u32x4*>(src + (((int64_t)b * H + y) * W + x) * C + ch * 8));
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], bf2f(v[e]));
                }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = f2bf(m[e]);
        }
        *reinterpret_cast<u32x4*>(out + row * C + ch * 8) = as_u32x4(o);
    }
}

// x: NHWC bf16 [B, P, C] (P = 49 positions).  att f32 [B, C, P]; fc f32 [B, C]; tokens bf16 [B*P, C] (optional copy)
__global__ void features_out_kernel(const bf16_t* __restrict__ x, float* __restrict__ att, float* __restrict__ fc,
                                    bf16_t* __restrict__ tokens, int B, int P, int C) {
    const int64_t total = (int64_t)B * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / C), c = (int)(i - (int64_t)b * C);
        float s = 0.f;
        for (int p = 0; p < P; ++p) {
            const bf16_t v = x[((int64_t)b * P + p) * C + c];
            const float f = bf2f(v);
            s += f;
            att[((int64_t)b * C + c) * P + p] = f;
            if (tokens) tokens[((int64_t)b * P + p) * C + c] = v;
        }
        fc[i] = s / (float)P;
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int icka_conv_stem_patches(const float* image, void* patches, int32_t B, int32_t H, int32_t W,
                                      int64_t rows_padded, void* stream) {
    if (!image || !patches) return ICKA_E_ARG;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    if (B <= 0 || H < 7 || W < 7 || rows_padded < (int64_t)B * Ho * Wo) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(stem_patches_kernel, dim3(grid_for(rows_padded * 24)), dim3(256), 0, (hipStream_t)stream, image,
                       (bf16_t*)patches, B, H, W, Ho, Wo, rows_padded);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_conv_im2col3x3(const void* src, void* patches, int32_t B, int32_t H, int32_t W, int32_t C,
                                   int32_t stride, int64_t rows_padded, void* stream) {
    if (!src || !patches) return ICKA_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (stride != 1 && stride != 2)) return ICKA_E_SHAPE;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    if (rows_padded < (int64_t)B * Ho * Wo) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(grid_for(rows_padded * 9 * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)patches, B, H, W, C, stride, Ho, Wo, rows_padded);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_conv_subsample(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                                   int64_t rows_padded, void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || stride < 1) return ICKA_E_SHAPE;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    if (rows_padded < (int64_t)B * Ho * Wo) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(subsample_kernel, dim3(grid_for(rows_padded * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, B, H, W, C, stride, Ho, Wo, rows_padded);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_conv_maxpool3x3s2(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C,
                                      int64_t rows_padded, void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8) return ICKA_E_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (rows_padded < (int64_t)B * Ho * Wo) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(rows_padded * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dst, B, H, W, C, Ho, Wo, rows_padded);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_conv_features_out(const void* x, float* att, float* fc, void* tokens, int32_t B, int32_t P, int32_t C,
                                      void* stream) {
    if (!x || !att || !fc) return ICKA_E_ARG;
    if (B <= 0 || P <= 0 || C <= 0) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(features_out_kernel, dim3(grid_for((int64_t)B * C)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, att, fc, (bf16_t*)tokens, B, P, C);
    ICKA_CHECK_LAUNCH();
    return 0;
}
