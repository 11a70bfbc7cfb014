// Replica of the GEMM compute-wave pipeline (no global traffic): per k-tile 16 ds_read_b128 + 32 MFMA with fragment
// prefetch distance 2, one s_barrier per k-tile, 4 compute waves (+4 barrier-only waves in MODE 1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE, int MF>   // MF: 16 -> 16x16x32 (16 acc tiles), 32 -> 32x32x16
__global__ __launch_bounds__(512) void pipe(unsigned long long* out, int iters) {
    __shared__ __attribute__((aligned(16))) char smem[98304];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 98304 / 4; i += blockDim.x) ((float*)smem)[i] = 1.0f;
    __syncthreads();
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        bf16x8 pa[4], pb[4], qa[4], qb[4], ra[4], rb[4], sa[4], sb[4];
        const char* base = smem + lane * 16 + (wave & 1) * 4096;
#define RD(FA, FB, OFF) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) { FA[i] = *(const bf16x8*)(base + (OFF) + i * 2048); FB[i] = *(const bf16x8*)(base + (OFF) + 16384 + i * 2048); } SB(); } while (0)
#define MM(FA, FB) do { _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[j], FA[i], acc[i * 4 + j], 0, 0, 0); SB(); } while (0)
        RD(pa, pb, 0); RD(qa, qb, 1024);
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            const int off = (it % 3) * 32768;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE >= 1) __builtin_amdgcn_s_barrier();
            RD(ra, rb, off);
            MM(pa, pb);
            RD(sa, sb, off + 1024);
            MM(qa, qb);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE >= 1) __builtin_amdgcn_s_barrier();
            RD(pa, pb, off);
            MM(ra, rb);
            RD(qa, qb, off + 1024);
            MM(sa, sb);
        }
        t1 = __builtin_amdgcn_s_memtime();
    } else if (MODE >= 1) {
        for (int it = 0; it < iters; ++it) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[1000] = 1;
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, unsigned long long* dout, int iters) {
    pipe<MODE, 16><<<256, threads>>>(dout, iters);
    pipe<MODE, 16><<<256, threads>>>(dout, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    (void)hipMemcpy(h.data(), dout, 256 * 8, hipMemcpyDeviceToHost);
    double m = 0;
    for (auto v : h) m += v;
    m /= 256;
    printf("%-60s %7.1f cycles per k-tile (32 MFMA + 16 b128; MFMA floor 544)\n", name, m / iters / 2);
}

int main() {
    unsigned long long* dout;
    (void)hipMalloc(&dout, 2048 * 8);
    run<0>("4 waves, no barrier", 256, dout, 1000);
    run<1>("4 compute + 4 barrier-only waves, barrier per k-tile", 512, dout, 1000);
    return 0;
}
