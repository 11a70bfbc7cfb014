// Micro-probes on one CU-sized launch: cycles per v_mfma_f32_16x16x32_bf16 alone, with ds_read_b128 / tr reads between,
// and with a co-resident wave issuing global_load_lds.  hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, const __bf16* gsrc, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[98304];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 98304 / 4; i += blockDim.x) ((float*)smem)[i] = 1.0f;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = *(bf16x8*)(smem + lane * 16 + i * 1024); b[i] = *(bf16x8*)(smem + 8192 + lane * 16 + i * 1024); }
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            if (MODE == 1 || MODE == 3) {   // 8 ds_read_b128 per 16 MFMA
                for (int i = 0; i < 4; ++i) { a[i] = *(volatile bf16x8*)(smem + ((it & 3) * 16384) + lane * 16 + i * 1024);
                                              b[i] = *(volatile bf16x8*)(smem + ((it & 3) * 16384) + 8192 + lane * 16 + i * 1024); }
            }
            if (MODE == 2) {   // 16 tr reads per 16 MFMA
                for (int i = 0; i < 4; ++i) {
                    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(smem + (it & 3) * 16384 + lane * 8 + i * 1024));
                    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(smem + (it & 3) * 16384 + 512 + lane * 8 + i * 1024));
                    a[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    bf16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(smem + 8192 + (it & 3) * 16384 + lane * 8 + i * 1024));
                    bf16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(smem + 8192 + (it & 3) * 16384 + 512 + lane * 8 + i * 1024));
                    b[i] = __builtin_shufflevector(lo2, hi2, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i * 4 + j], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
    } else if (MODE == 3) {   // loader waves: 8 global_load_lds per iteration, like the GEMM ring
        const __bf16* p = gsrc + (size_t)(blockIdx.x * 4 + (wave - 4)) * 65536 + lane * 8;
        for (int it = 0; it < iters; ++it) {
            for (int j = 0; j < 8; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p + j * 512 + (it & 15) * 4096),
                                                 (__attribute__((address_space(3))) void*)(smem + 65536 + (wave - 4) * 8192 + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[1000] = 1;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, unsigned long long* dout, __bf16* gsrc, int iters) {
    probe<MODE><<<256, threads>>>(dout, gsrc, iters);
    probe<MODE><<<256, threads>>>(dout, gsrc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += v;
    m /= 256;
    printf("%-46s %7.1f cycles per 16 MFMA (ideal 256), %5.1f per MFMA\n", name, m / iters, m / iters / 16);
}

int main() {
    unsigned long long* dout;
    __bf16* gsrc;
    hipMalloc(&dout, 2048 * 8);
    hipMalloc(&gsrc, (size_t)256 * 4 * 65536 * 2 + (1 << 20));
    hipMemset(gsrc, 0, (size_t)256 * 4 * 65536 * 2 + (1 << 20));
    const int iters = 2000;
    run<0>("bare 16 MFMA, 1 wave/SIMD", 256, dout, gsrc, iters);
    run<1>("8 ds_read_b128 + 16 MFMA, 1 wave/SIMD", 256, dout, gsrc, iters);
    run<2>("16 ds_read_b64_tr_b16 + 16 MFMA, 1 wave/SIMD", 256, dout, gsrc, iters);
    run<3>("8 b128 + 16 MFMA beside a loader wave (8 glds/it)", 512, dout, gsrc, iters);
    run<1>("8 ds_read_b128 + 16 MFMA, 2 compute waves/SIMD?", 512, dout, gsrc, iters);
    return 0;
}
