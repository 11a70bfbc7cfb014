// Data-parallel helpers (reference: apex DistributedDataParallel, My_cross_attention.py:768-776).
//
// The gradient exchange itself is RCCL (torch.distributed, backend "nccl") over xGMI; this file holds what runs around it on
// the GPU so that the exchange costs the compute stream as little as possible:
//
//   * wire format.  Gradients travel as bf16.  The weight-gradient GEMMs write the bf16 copy of every matrix gradient into the
//     wire buffer from their own epilogue (icka_gemm_desc.C3, gemm.hip), so only what no GEMM produces -- bias / LayerNorm
//     vectors, the classifier, the embedding tables -- still has to be cast: icka_dp_cast_chunks does that for one bucket in
//     ONE launch over a chunk table (scattered ranges of the flat gradient buffer), and icka_dp_cast_back_scaled brings the
//     reduced bucket back to f32 with the 1/world factor folded in (the all-reduce is then a plain SUM).
//   * bucket-ready flags.  The step is ONE captured hipGraph; the point where a gradient bucket becomes final is a tiny node
//     of that graph (icka_dp_flag_set) that publishes the step number in the bucket's flag word.  The all-reduces are NOT
//     captured: per bucket the host enqueues, on the communication stream, a one-wave kernel that waits for the flag
//     (icka_dp_flag_wait; bounded spin with s_sleep) ahead of the eager collective.  No graph cut, no cross-stream event
//     inside the graph, no captured collective.  A wait that gives up raises a host-visible error word (host memory mapped
//     into the device: the host polls it without a synchronisation) and stores the step number into the bucket's BAD WORD
//     (device memory).  The poison is applied by kernels that RUN AFTER whatever could overwrite it: icka_dp_poison_if on the
//     communication stream between the bucket's chunk cast and its all-reduce (the NaN travels to every rank) and again after
//     the cast-back, and icka_dp_poison_final on the compute stream after the join at the end of the step, when the graph's
//     late gradient stores are over -- so an all-reduce of unfinished gradients can never pass for a result (round 3 wrote the
//     NaN from the wait kernel itself, and the chunk cast / a late GEMM epilogue overwrote it: ADVICE r03).
#include "common.h"

namespace {

constexpr int DP_CHUNK = 8192;   // elements per chunk-table entry (and per block)

// table[2 b] = first element, table[2 b + 1] = elements (multiple of 8, <= DP_CHUNK) of chunk b
__global__ __launch_bounds__(256) void dp_cast_chunks_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             const int64_t* __restrict__ table) {
    const int64_t lo = table[2 * blockIdx.x], len = table[2 * blockIdx.x + 1];
    const float* s = src + lo;
    bf16_t* d = dst + lo;
    for (int64_t c = threadIdx.x * 8; c < len; c += 256 * 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(s + c), b = *reinterpret_cast<const f32x4*>(s + c + 4);
        bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
        *reinterpret_cast<u32x4*>(d + c) = as_u32x4(o);
    }
}

__global__ __launch_bounds__(256) void dp_cast_back_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, int64_t n,
                                                           float scale) {
    const int64_t nch = n >> 3;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < nch; c += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = as_bf16x8(__builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + c * 8)));
        *reinterpret_cast<f32x4*>(dst + c * 8) = f32x4{bf2f(v[0]) * scale, bf2f(v[1]) * scale, bf2f(v[2]) * scale, bf2f(v[3]) * scale};
        *reinterpret_cast<f32x4*>(dst + c * 8 + 4) = f32x4{bf2f(v[4]) * scale, bf2f(v[5]) * scale, bf2f(v[6]) * scale, bf2f(v[7]) * scale};
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[(nch << 3) + threadIdx.x] = bf2f(src[(nch << 3) + threadIdx.x]) * scale;
}

__global__ void dp_step_bump_kernel(unsigned int* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] = step[0] + 1u;
}
// a node of the step's graph, placed right after the kernel that makes the bucket's last gradient final: stream order puts
// every earlier kernel's writes (released at their kernel boundaries) before this store
__global__ void dp_flag_set_kernel(unsigned int* flag, const unsigned int* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, step[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one wave on the communication stream, ahead of the bucket's all-reduce: returns once flag >= tag (wrap-safe)
__global__ void dp_flag_wait_kernel(const unsigned int* flag, unsigned int tag, unsigned int* err, unsigned int* bad,
                                    int max_polls) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int polls = 0;
    while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - tag) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if (++polls > max_polls) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (bad) __hip_atomic_store(bad, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
// one wave: NaN into the first n (<= 64) elements of a bucket whose wait gave up in step `tag`
__global__ void dp_poison_if_kernel(const unsigned int* bad, unsigned int tag, void* target, int is_bf16, int n) {
    if (blockIdx.x != 0 || (int)threadIdx.x >= n) return;
    if (__hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) return;
    if (is_bf16) reinterpret_cast<unsigned short*>(target)[threadIdx.x] = (unsigned short)0x7fc0;
    else reinterpret_cast<unsigned int*>(target)[threadIdx.x] = 0x7fc00000u;
}
// end of the step, compute stream: every bucket whose bad word carries this step's number gets NaN in the first n elements of
// its f32 gradients (starts[b] = first element of bucket b in the flat gradient buffer)
__global__ void dp_poison_final_kernel(const unsigned int* bad, int n_buckets, unsigned int tag, float* gflat,
                                       const int64_t* starts, int n) {
    const int b = blockIdx.x;
    if (b >= n_buckets || (int)threadIdx.x >= n) return;
    if (__hip_atomic_load(bad + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != tag) return;
    reinterpret_cast<unsigned int*>(gflat + starts[b])[threadIdx.x] = 0x7fc00000u;
}

volatile unsigned int* g_dp_err_host = nullptr;
unsigned int* g_dp_err_dev = nullptr;

}  // namespace

extern "C" int64_t icka_dp_chunk_elems(void) { return DP_CHUNK; }

extern "C" int icka_dp_cast_chunks(const float* src, void* dst, const int64_t* table_dev, int32_t n_chunks, void* stream) {
    if (!src || !dst || (n_chunks > 0 && !table_dev)) return ICKA_E_ARG;
    if (n_chunks <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(src) & 31) || (reinterpret_cast<uintptr_t>(dst) & 15)) return ICKA_E_ALIGN;
    hipLaunchKernelGGL(dp_cast_chunks_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, table_dev);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_dp_cast_back_scaled(const void* src, float* dst, int64_t n, float scale, void* stream) {
    if (!src || !dst) return ICKA_E_ARG;
    if (n <= 0) return 0;
    if ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return ICKA_E_ALIGN;
    int64_t blocks = ((n + 7) / 8 + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(dp_cast_back_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, dst, n,
                       scale);
    ICKA_CHECK_LAUNCH();
    return 0;
}

/* Maps the host-visible error word (once; never call inside a stream capture). */
extern "C" int icka_dp_init(void) {
    if (g_dp_err_dev) return 0;
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        return ICKA_E_ARG;
    }
    *reinterpret_cast<volatile unsigned int*>(h) = 0u;
    g_dp_err_host = reinterpret_cast<volatile unsigned int*>(h);
    g_dp_err_dev = reinterpret_cast<unsigned int*>(d);
    return 0;
}
extern "C" int icka_dp_error(void) { return g_dp_err_host ? (int)*g_dp_err_host : 0; }
extern "C" int icka_dp_clear_error(void) {
    if (g_dp_err_host) *g_dp_err_host = 0u;
    return 0;
}

extern "C" int icka_dp_step_bump(void* step_word, void* stream) {
    if (!step_word) return ICKA_E_ARG;
    hipLaunchKernelGGL(dp_step_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned int*)step_word);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dp_flag_set(void* flag_word, const void* step_word, void* stream) {
    if (!flag_word || !step_word) return ICKA_E_ARG;
    hipLaunchKernelGGL(dp_flag_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned int*)flag_word,
                       (const unsigned int*)step_word);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dp_flag_wait(const void* flag_word, uint32_t tag, void* bad_word, int32_t max_polls, void* stream) {
    if (!flag_word || max_polls <= 0) return ICKA_E_ARG;
    if (!g_dp_err_dev && icka_dp_init() != 0) return ICKA_E_ARG;
    hipLaunchKernelGGL(dp_flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned int*)flag_word, tag,
                       g_dp_err_dev, (unsigned int*)bad_word, max_polls);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dp_poison_if(const void* bad_word, uint32_t tag, void* target, int32_t target_is_bf16, int32_t n,
                                 void* stream) {
    if (!bad_word || !target || n <= 0 || n > 64) return ICKA_E_ARG;
    hipLaunchKernelGGL(dp_poison_if_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const unsigned int*)bad_word, tag, target,
                       target_is_bf16, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
extern "C" int icka_dp_poison_final(const void* bad_words, int32_t n_buckets, uint32_t tag, float* gflat,
                                    const int64_t* starts_dev, int32_t n, void* stream) {
    if (!bad_words || !gflat || !starts_dev || n_buckets <= 0 || n <= 0 || n > 64) return ICKA_E_ARG;
    hipLaunchKernelGGL(dp_poison_final_kernel, dim3(n_buckets), dim3(64), 0, (hipStream_t)stream, (const unsigned int*)bad_words,
                       n_buckets, tag, gflat, starts_dev, n);
    ICKA_CHECK_LAUNCH();
    return 0;
}
