// XCD-local producer -> consumer seam: what does it cost to hand a GEMM output stripe to a LayerNorm phase
// INSIDE one launch when producers and consumers are the CUs of ONE XCD (shared 4 MiB L2), against a launch boundary?
//
// Emulates the out-proj GEMM -> LayerNorm seam of a BERT layer at c2 (4096 x 768 bf16): 256 blocks (one per CU); block b
// belongs to XCD b & 7 (round-robin dispatch: checked against HW_REG_XCC_ID and reported), stripe (b >> 3) >> 3 of its XCD,
// column tile (b >> 3) & 7.  Phase 1: the block writes its 128 x 96 bf16 tile of C (24 KiB) -- what the GEMM epilogue stores.
// Seam: one arrival per block on the stripe's counter; the 8 blocks of a stripe wait for all 8.  Phase 2: block j of the stripe
// reads rows 16 j .. 16 j + 15 of the stripe (all 768 columns: 24 KiB produced by the 8 blocks) and checks every element.
//   MODE 0: plain stores, agent-scope release fence, relaxed atomic add; consumer: relaxed poll, agent acquire fence, plain loads
//   MODE 1: write-through (sc1) stores, s_waitcnt vmcnt(0), atomic add;   consumer: sc1 poll, sc1 loads (no fence)
// and, for comparison, the same two phases as TWO launches (the launch boundary is the seam).
// Every wait is bounded (a give-up is counted and reported); every element is verified (stale reads are counted).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/xcd_seam_probe.hip -o tools/probe/xcd_seam_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int M = 4096, N = 768, TM = 128, TN = 96, NT = N / TN;   // 8 column tiles per stripe
constexpr int MAX_POLLS = 1 << 18;

__device__ __forceinline__ unsigned pattern(int row, int col8, int iter) {   // value of the 8-element group (row, col8): 16 bytes
    return (unsigned)(row * 131 + col8 * 7 + iter * 1000003);
}
__device__ __forceinline__ void store16_sc1(void* p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }   // (s_nop: store-data hazard)
__device__ __forceinline__ u32x4 load16_sc1(const void* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

struct Args {
    unsigned short* C;          // [M][N] bf16
    unsigned int* counters;     // [iters][32 stripes * 16] arrival counters (one 64-byte line each)
    unsigned long long* stamps; // [256][8]
    unsigned int* errs;         // [0] stale elements, [1] give-ups, [2] blocks whose XCC_ID != (b & 7) relative to block 0's
    unsigned int* xcc;          // [256] XCC_ID per block
    int iter;                   // index of the arrival counters of this launch
    int tag;                    // unique per launch over the whole run: folded into the data pattern (stale reads are detected)
};

__device__ __forceinline__ void produce(const Args& a, int stripe_g, int j, int mode) {
    // 128 rows x 96 columns = 128 x 12 groups of 16 bytes: 1536 groups over 256 threads = 6 each
    for (int i = threadIdx.x; i < TM * (TN / 8); i += 256) {
        const int r = i / (TN / 8), c8 = i % (TN / 8);
        const int row = stripe_g * TM + r, col8 = j * (TN / 8) + c8;
        const unsigned v = pattern(row, col8, a.tag);
        u32x4 w = {v, v + 1, v + 2, v + 3};
        void* p = a.C + (size_t)row * N + col8 * 8;
        if (mode == 1) store16_sc1(p, w); else *reinterpret_cast<u32x4*>(p) = w;
    }
}
__device__ __forceinline__ unsigned consume(const Args& a, int stripe_g, int j, int mode) {
    unsigned bad = 0;
    // rows 16 j .. 16 j + 15 of the stripe, 96 groups per row: 1536 groups over 256 threads
    for (int i = threadIdx.x; i < 16 * (N / 8); i += 256) {
        const int r = i / (N / 8), col8 = i % (N / 8);
        const int row = stripe_g * TM + 16 * j + r;
        const void* p = a.C + (size_t)row * N + col8 * 8;
        const u32x4 w = mode == 1 ? load16_sc1(p) : *reinterpret_cast<const u32x4*>(p);
        const unsigned v = pattern(row, col8, a.tag);
        bad += (w[0] != v) + (w[1] != v + 1) + (w[2] != v + 2) + (w[3] != v + 3);
    }
    return bad;
}

template <int MODE>
__global__ __launch_bounds__(256) void fused(Args a) {
    __shared__ char pad[96 * 1024];      // one block per CU
    __shared__ unsigned s_ok;
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3, stripe = li >> 3, j = li & 7;
    const int stripe_g = xcd * 4 + stripe;                                   // 32 stripes of 128 rows
    if (threadIdx.x == 0) { a.xcc[b] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) ; pad[0] = 0; }   // HW_REG_XCC_ID[3:0]
    const unsigned long long t0 = rt();
    produce(a, stripe_g, j, MODE);
    unsigned int* ctr = a.counters + ((size_t)a.iter * 32 + stripe_g) * 16;
    if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                   // buffer_wbl2 sc1 + wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // write-through stores have left the CU
    }
    __syncthreads();
    const unsigned long long t1 = rt();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 0;
        for (int p = 0; p < MAX_POLLS; ++p) {
            if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 8u) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        s_ok = ok;
        if (!ok) atomicAdd(a.errs + 1, 1u);
    }
    __syncthreads();
    const unsigned long long t2 = rt();
    if (MODE == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                   // buffer_inv sc1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const unsigned bad = s_ok ? consume(a, stripe_g, j, MODE) : 0;
    if (bad) atomicAdd(a.errs, bad);
    __syncthreads();
    const unsigned long long t3 = rt();
    if (threadIdx.x == 0) {
        unsigned long long* s = a.stamps + (size_t)b * 8;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3;
    }
}

__global__ __launch_bounds__(256) void producer_only(Args a) {
    __shared__ char pad[96 * 1024];
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
    if (threadIdx.x == 0) pad[0] = 0;
    const unsigned long long t0 = rt();
    produce(a, xcd * 4 + (li >> 3), li & 7, 0);
    if (threadIdx.x == 0) a.stamps[(size_t)b * 8] = t0;
}
__global__ __launch_bounds__(256) void consumer_only(Args a) {
    __shared__ char pad[96 * 1024];
    const int b = blockIdx.x, xcd = b & 7, li = b >> 3;
    if (threadIdx.x == 0) pad[0] = 0;
    const unsigned bad = consume(a, xcd * 4 + (li >> 3), li & 7, 0);
    if (bad) atomicAdd(a.errs, bad);
    __syncthreads();
    if (threadIdx.x == 0) a.stamps[(size_t)b * 8 + 3] = rt();
}

static void report(const char* name, const std::vector<unsigned long long>& st, int iters, double event_us) {
    // stamps of the LAST iteration: ticks of 10 ns
    double pub = 0, wait = 0, rd = 0, wmax = 0;
    unsigned long long first = ~0ull, last = 0;
    for (int b = 0; b < 256; ++b) {
        const unsigned long long* s = &st[(size_t)b * 8];
        pub += (s[1] - s[0]) * 0.01; wait += (s[2] - s[1]) * 0.01; rd += (s[3] - s[2]) * 0.01;
        wmax = std::max(wmax, (s[2] - s[1]) * 0.01);
        first = std::min(first, s[0]); last = std::max(last, s[3]);
    }
    printf("%-44s per block (us): store+publish %5.2f  wait for the stripe %5.2f (max %5.2f)  acquire+read %5.2f | grid span %6.2f us | "
           "%6.2f us per launch over %d back-to-back launches (HIP events)\n", name, pub / 256, wait / 256, wmax, rd / 256,
           (last - first) * 0.01, event_us, iters);
}

int main() {
    Args a{};
    const int iters = 50;
    CHECK(hipMalloc(&a.C, (size_t)M * N * 2));
    CHECK(hipMalloc(&a.counters, (size_t)(iters + 1) * 3 * 32 * 16 * 4));
    CHECK(hipMalloc(&a.stamps, 256 * 8 * 8));
    CHECK(hipMalloc(&a.errs, 16));
    CHECK(hipMalloc(&a.xcc, 256 * 4));
    CHECK(hipMemset(a.counters, 0, (size_t)(iters + 1) * 3 * 32 * 16 * 4));
    CHECK(hipMemset(a.errs, 0, 16));
    CHECK(hipMemset(a.C, 0, (size_t)M * N * 2));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<unsigned long long> st(256 * 8);
    unsigned int* ctr0 = a.counters;
    int it_global = 0;
    for (int mode = 0; mode < 3; ++mode) {
        a.counters = ctr0 + (size_t)mode * (iters + 1) * 32 * 16;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {       // rep 0 warms up
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < (rep ? iters : 1); ++it) {
                a.iter = rep ? it + 1 : 0;
                Args k = a;
                k.tag = it_global;
                if (mode == 0) hipLaunchKernelGGL(fused<0>, dim3(256), dim3(256), 0, 0, k);
                else if (mode == 1) hipLaunchKernelGGL(fused<1>, dim3(256), dim3(256), 0, 0, k);
                else { hipLaunchKernelGGL(producer_only, dim3(256), dim3(256), 0, 0, k); hipLaunchKernelGGL(consumer_only, dim3(256), dim3(256), 0, 0, k); }
                ++it_global;
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        CHECK(hipMemcpy(st.data(), a.stamps, st.size() * 8, hipMemcpyDeviceToHost));
        if (mode == 2) {
            unsigned long long first = ~0ull, last = 0;
            for (int b = 0; b < 256; ++b) { first = std::min(first, st[(size_t)b * 8]); last = std::max(last, st[(size_t)b * 8 + 3]); }
            printf("%-44s first producer block start -> last consumer block end %6.2f us | %6.2f us per PAIR of launches over %d pairs\n",
                   "two launches (launch boundary as the seam)", (last - first) * 0.01, ms * 1e3 / iters, iters);
        } else {
            report(mode == 0 ? "fused, plain stores + release / acquire" : "fused, sc1 stores + sc1 loads (no fences)", st, iters,
                   ms * 1e3 / iters);
        }
    }
    unsigned errs[4] = {0, 0, 0, 0};
    std::vector<unsigned> xcc(256);
    CHECK(hipMemcpy(errs, a.errs, 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(xcc.data(), a.xcc, 256 * 4, hipMemcpyDeviceToHost));
    int mis = 0;
    for (int b = 0; b < 256; ++b) mis += ((xcc[b] & 15) != ((xcc[0] + (b & 7)) & 7)) ? 1 : 0;
    printf("stale elements read: %u   waits that gave up: %u   blocks not on XCD (xcc[0] + b) %% 8: %d of 256 (xcc of blocks 0..7: %u %u %u %u %u %u %u %u)\n",
           errs[0], errs[1], mis, xcc[0] & 15, xcc[1] & 15, xcc[2] & 15, xcc[3] & 15, xcc[4] & 15, xcc[5] & 15, xcc[6] & 15, xcc[7] & 15);
    return (errs[0] || errs[1]) ? 2 : 0;
}
