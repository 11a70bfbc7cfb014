// What does the L2 -> L1 -> LDS path of a CU deliver when the CUs of an XCD ask for the SAME lines (GEMM operand panels) rather
// than private ones?  256 blocks x W loader waves stream L2-resident regions with global_load_lds_dwordx4, DEPTH 1-KiB
// wave-loads in flight per wave, in the GEMM's piece shape (8 rows x 128 B, rows 6 KiB apart, 128-row panels):
//   share 0  every block its own panel            share 1  the 32 blocks of an XCD (blockIdx % 8) read one panel in lockstep
//   share 2  GEMM-like: 16 KiB of each 28 KiB k-tile from a panel shared by 8 blocks, 12 KiB from one shared by 32
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/share_probe.hip -o /tmp/share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); exit(1); } } while (0)
template <int N> __device__ __forceinline__ void waitvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int ROWB = 6144;         // bytes between rows (K = 3072 bf16)
// ksl = 128-byte k-slices used of each row: 4 -> 64 KiB per panel (L2 resident), 48 -> the whole K = 3072 row: 768 KiB per
// panel, 40 panels = 30 MiB in the GEMM-like case (the real operand footprint: streams through the L2s from the Infinity Cache)
constexpr size_t PANEL = (size_t)128 * ROWB;

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream(const char* base, int passes, int nwaves, int share, int KSL, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) char smem[16 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
    const char* pa;   // panel of the "A" pieces
    const char* pb;   // panel of the "B" pieces
    if (share == 0) { pa = pb = base + (size_t)b * PANEL; }
    else if (share == 1) { pa = pb = base + (size_t)xcd * PANEL; }
    else { pa = base + (size_t)(xcd * 4 + (li >> 3)) * PANEL; pb = base + (size_t)(32 + xcd) * PANEL; }
    const unsigned lds = (unsigned)(size_t)(smem + wave * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int inflight = 0;
    // one "k-tile" = 28 pieces (16 of A: 128 rows, 12 of B: 96 rows) of the 128-byte k-slice kt; GEMM order
    for (int pass = 0; pass < passes; ++pass)
        for (int kt = 0; kt < KSL; ++kt)
            for (int p = wave; p < 28; p += nwaves) {
                const char* pan = p < 16 ? pa : pb;
                const int rp = p < 16 ? p : p - 16;
                const char* a = pan + (size_t)(rp * 8 + (lane >> 3)) * ROWB + (size_t)kt * 128 + (lane & 7) * 16;
                if (inflight == DEPTH) { waitvm<DEPTH - 1>(); inflight = DEPTH - 1; }
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(lds) : "memory");
                ++inflight;
            }
    waitvm<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

// the GEMM loader's own issue shape without its barrier: per k-tile a wave issues ITS pieces (28 / nwaves) back to back into
// a ring slot, then waits until at most AHEAD younger k-tiles of its own are still in flight (one counted vmcnt per k-tile)
template <int PER, int AHEAD>
__global__ __launch_bounds__(1024) void burst(const char* base, int passes, int nwaves, int share, int KSL, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 28 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
    const char* pa;
    const char* pb;
    if (share == 0) { pa = pb = base + (size_t)b * PANEL; }
    else if (share == 1) { pa = pb = base + (size_t)xcd * PANEL; }
    else { pa = base + (size_t)(xcd * 4 + (li >> 3)) * PANEL; pb = base + (size_t)(32 + xcd) * PANEL; }
    const unsigned lds0 = (unsigned)(size_t)smem;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int slot = 0;
    for (int pass = 0; pass < passes; ++pass)
        for (int kt = 0; kt < KSL; ++kt) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const int p = wave + nwaves * j;
                const char* pan = p < 16 ? pa : pb;
                const int rp = p < 16 ? p : p - 16;
                const char* a = pan + (size_t)(rp * 8 + (lane >> 3)) * ROWB + (size_t)kt * 128 + (lane & 7) * 16;
                const unsigned lds = lds0 + slot * 28 * 1024 + p * 1024;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(lds) : "memory");
            }
            waitvm<PER * AHEAD>();
            slot = slot + 1 == AHEAD + 1 ? 0 : slot + 1;
        }
    waitvm<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
}

template <int PER, int AHEAD>
static void run_burst(const char* buf, int passes, int share, int KSL, unsigned long long* out) {
    const int nwaves = 28 / PER;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((burst<PER, AHEAD>), dim3(256), dim3(1024), 0, 0, buf, passes, nwaves, share, KSL, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((burst<PER, AHEAD>), dim3(256), dim3(1024), 0, 0, buf, passes, nwaves, share, KSL, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)passes * KSL * 28 * 1024.0;
    printf("burst issue, GEMM-like sharing     ksl %2d waves %2d x %d pieces, %d k-tiles ahead : %6.2f TB/s aggregate  %6.1f ns per 28-KiB k-tile\n",
           KSL, nwaves, PER, AHEAD, bytes * 256 / (ms * 1e-3) * 1e-12, ms * 1e6 / (passes * KSL));
}

template <int DEPTH>
static void run(const char* buf, int passes, int nwaves, int share, int KSL, unsigned long long* out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((stream<DEPTH>), dim3(256), dim3(1024), 0, 0, buf, passes, nwaves, share, KSL, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((stream<DEPTH>), dim3(256), dim3(1024), 0, 0, buf, passes, nwaves, share, KSL, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256];
    CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
    double cyc = 0;
    for (int i = 0; i < 256; ++i) cyc += (double)h[i];
    cyc /= 256;
    const double bytes = (double)passes * KSL * 28 * 1024.0;
    const char* names[3] = {"private panels", "one panel per XCD (32 sharers)", "GEMM-like (A x8, B x32 sharers)"};
    printf("%-34s ksl %2d waves %2d depth %2d : %6.1f B/memclk/CU  %6.2f TB/s aggregate  %6.1f ns per 28-KiB k-tile  (%.1f us)\n",
           names[share], KSL, nwaves, DEPTH, bytes / cyc, bytes * 256 / (ms * 1e-3) * 1e-12, ms * 1e6 / (passes * KSL), ms * 1e3);
}

int main() {
    const size_t total = (size_t)256 * PANEL;   // 192 MiB of address space; touched: 64 KiB per panel
    char* buf;
    unsigned long long* out;
    CK(hipMalloc(&buf, total)); CK(hipMalloc(&out, 2048 * 8));
    CK(hipMemset(buf, 1, total));
    for (int ksl : {4, 48})
        for (int share = 0; share < 3; ++share)
            for (int nw : {4, 8, 16}) {
                if (ksl == 48 && share == 0) continue;   // 192 MiB of private panels: not the GEMM's situation
                run<8>(buf, 256 / ksl, nw, share, ksl, out);
                run<16>(buf, 256 / ksl, nw, share, ksl, out);
            }
    // the loader's burst shape: 4 waves x 7 pieces (the kernel's), 7 x 4, 14 x 2, with 1 .. 4 k-tiles in flight behind the newest
    run_burst<7, 1>(buf, 6, 2, 48, out); run_burst<7, 2>(buf, 6, 2, 48, out); run_burst<7, 3>(buf, 6, 2, 48, out); run_burst<7, 4>(buf, 6, 2, 48, out);
    run_burst<4, 1>(buf, 6, 2, 48, out); run_burst<4, 2>(buf, 6, 2, 48, out); run_burst<4, 3>(buf, 6, 2, 48, out); run_burst<4, 4>(buf, 6, 2, 48, out);
    run_burst<2, 1>(buf, 6, 2, 48, out); run_burst<2, 2>(buf, 6, 2, 48, out); run_burst<2, 3>(buf, 6, 2, 48, out); run_burst<2, 4>(buf, 6, 2, 48, out);
    return 0;
}
