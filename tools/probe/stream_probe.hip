// How fast can ONE CU pull data through L1 -> LDS (or -> VGPRs)?  256 blocks (one per CU) stream private regions with W
// loader waves keeping DEPTH 1-KiB wave-loads in flight each; no compute, no LDS reads.  Reports bytes per shader clock
// per CU (s_memtime of wave 0) and the aggregate rate (HIP events), for
//   path    lds-dma  global_load_lds_dwordx4 (the GEMM operand path)      | vgpr  global_load_dwordx4
//   data    cold     a 2 GiB footprint walked once                         | warm  64 KiB per block, re-read (L2 resident)
//   pattern seq      consecutive 1-KiB pieces                              | tile  8 rows x 128 B per piece, rows 6 KiB apart
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/stream_probe.hip -o /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int N> __device__ __forceinline__ void waitvm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int DEPTH>
__global__ __launch_bounds__(1024) void stream(const char* base, size_t per_block, int pieces, int passes, int nwaves,
                                               int pattern, unsigned long long* out) {
    __shared__ __attribute__((aligned(16))) char smem[16 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= nwaves) return;
    const char* p0 = base + (size_t)blockIdx.x * per_block;
    const unsigned lds = (unsigned)(size_t)(smem + wave * 1024);
    u32x4 sink = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int inflight = 0;
    for (int pass = 0; pass < passes; ++pass) {
        for (int p = wave; p < pieces; p += nwaves) {
            const char* a;
            if (pattern == 0) a = p0 + (size_t)p * 1024 + lane * 16;
            else {   // GEMM-like: piece = 8 rows of one 128-byte k-slice; 16 pieces (128 rows) per k-slice
                const int kt = p >> 4, rp = p & 15;
                a = p0 + (size_t)(rp * 8 + (lane >> 3)) * 6144 + (size_t)kt * 128 + (lane & 7) * 16;
            }
            if (inflight == DEPTH) { waitvm<DEPTH - 1>(); inflight = DEPTH - 1; }
            if (MODE == 0) {
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(lds) : "memory");
            } else {
                u32x4 v;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(a) : "memory");
                sink ^= v;   // (no wait: the compiler cannot see the load, the value is junk until vmcnt drains -- unused)
            }
            ++inflight;
        }
    }
    waitvm<0>();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) out[blockIdx.x] = t1 - t0;
    if (sink[0] == 0x12345678u && out) out[1024] = sink[1];
}

template <int MODE, int DEPTH>
static void run(const char* buf, size_t per_block, int pieces, int passes, int nwaves, int pattern, unsigned long long* out,
                const char* what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream<MODE, DEPTH>), dim3(256), dim3(1024), 0, 0, buf, per_block, pieces, passes, nwaves, pattern, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream<MODE, DEPTH>), dim3(256), dim3(1024), 0, 0, buf, per_block, pieces, passes, nwaves, pattern, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < 256; ++i) cyc += (double)h[i];
    cyc /= 256;
    const double bytes = (double)pieces * passes * 1024.0;
    printf("%-34s waves %2d depth %2d : %6.1f B/clk/CU   %6.2f TB/s aggregate   (%.0f cycles, %.1f us)\n", what, nwaves, DEPTH,
           bytes / cyc, bytes * 256 / (ms * 1e-3) * 1e-12, cyc, ms * 1e3);
}

int main() {
    const size_t total = (size_t)2 << 30;
    char* buf;
    unsigned long long* out;
    if (hipMalloc(&buf, total) != hipSuccess || hipMalloc(&out, 2048 * 8) != hipSuccess) return 1;
    hipMemset(buf, 1, total);
    const size_t cold_pb = total / 256;               // 8 MiB per block, walked once
    const int cold_pieces = 4096;                     // 4 MiB of it (seq) / 256 k-slices of a 128-row panel (tile)
    for (int pattern = 0; pattern < 2; ++pattern) {
        const char* pn = pattern ? "tile" : "seq ";
        char w[64];
        for (int nw : {4, 8, 16}) {
            snprintf(w, sizeof w, "lds-dma cold %s", pn); run<0, 8>(buf, cold_pb, cold_pieces, 1, nw, pattern, out, w);
            run<0, 16>(buf, cold_pb, cold_pieces, 1, nw, pattern, out, w);
            run<0, 32>(buf, cold_pb, cold_pieces, 1, nw, pattern, out, w);
        }
        snprintf(w, sizeof w, "vgpr    cold %s", pn);
        run<1, 8>(buf, cold_pb, cold_pieces, 1, 4, pattern, out, w);
        run<1, 16>(buf, cold_pb, cold_pieces, 1, 8, pattern, out, w);
        run<1, 32>(buf, cold_pb, cold_pieces, 1, 16, pattern, out, w);
    }
    // warm: 64 KiB per block re-read 64 times (16 MiB total: L2 / MALL resident after the first launch)
    for (int nw : {4, 8, 16}) {
        run<0, 8>(buf, 65536, 64, 64, nw, 0, out, "lds-dma warm seq");
        run<0, 32>(buf, 65536, 64, 64, nw, 0, out, "lds-dma warm seq");
    }
    run<1, 16>(buf, 65536, 64, 64, 8, 0, out, "vgpr    warm seq");
    return 0;
}
