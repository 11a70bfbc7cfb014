// Bidirectional single-layer LSTM of the tagging tail:
//   self.lstm = nn.LSTM(input_size=H, hidden_size=H, batch_first=True, bidirectional=True)
//   x, _ = self.lstm(result); emissions = self.classifier(x)      (Cross_Modal_Interaction_Module.py:905-910, :1042-1043)
// nn.LSTM is ATen, not reference code; semantics (gate order i, f, g, o; c' = f*c + i*g; h' = o*tanh(c')) follow
// torch.nn.LSTM.
//
// The input projection of all S steps and both directions is ONE GEMM of the GEMM kernels ([B*S, H] x [H, 8H], the
// reference hoists nothing: cuDNN/ATen does the same internally).  This file holds the sequential part: one launch per
// time step (both directions in the launch, captured in the step's hipGraph), a block per 16 hidden units:
//   forward : G = h_{t-1} . W_hh^T for its 4 x 16 gate columns on the MFMA (one gate per wave, operands straight from
//             L2 -- the 4H x H recurrent matrix is 4.7 MB and stays L2-resident), then the cell update for its units.
//   backward: dh_rec = dgates_{t+1} . W_hh for its 16 units (K = 4H split over the 4 waves), then the gate
//             derivatives of step t.  The weight / input gradients are GEMMs over all steps after the loop.
// The state lives in the saved tensors: h_{t-1} is read from the output y, c_{t-1} from c_all.
#include <math.h>

#include "common.h"

namespace {

struct LstmArgs {
    const float* gx; int64_t ldg;      // input projection + both biases, f32 [B*S, 8H]: col = dir*4H + gate*H + unit
    const bf16_t* whh;                 // forward: [2][4H][H] (gate-row, k contiguous); backward: W_hh^T [2][H][4H]
    bf16_t* y;                         // [B, S, 2H] hidden states (= LSTM output), col = dir*H + unit
    float* c_all;                      // [B, S, 2, H] cell states
    bf16_t* act;                       // [B, S, 2, 4H] gate activations i, f, g, o
    bf16_t* hprev;                     // [B, S, 2H] h_{t-1} of every step (operand of the dW_hh GEMM), may be NULL
    const bf16_t* dy;                  // [B, S, 2H] gradient of the output
    bf16_t* dgates;                    // [B*S, 8H] gate pre-activation gradients
    float* dc_carry;                   // [2, B, H] dc_{t+1} * f_{t+1}
    int B, S, H, step;
    int poll_limit;                    // persistent forms: polls before a hand-off wait gives up (error word + NaN poison)
    int test_drop;                     // test hook (icka_lstm_test_hooks): block (0, 0, 0) does not publish this step; -1 = off
};

// 16-byte fragment of a k-contiguous row (8 bf16); zero when the row is invalid
__device__ __forceinline__ bf16x8 frag16(const bf16_t* row, bool ok) {
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ok) v = *reinterpret_cast<const u32x4*>(row);
    return as_bf16x8(v);
}

constexpr int MAX_RT = 4;   // batch row tiles of 16 (B <= 64 per launch)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)__builtin_bit_cast(unsigned short, f2bf(lo)) | ((uint32_t)__builtin_bit_cast(unsigned short, f2bf(hi)) << 16);
}

// ---------------------------------------------------------------------------------------------------- forward step
template <int NRT, int UB>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const LstmArgs a) {
    __shared__ float s_g[4][MAX_RT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, u0 = blockIdx.x * 16, H = a.H, S = a.S;
    const int tt = d == 0 ? a.step : S - 1 - a.step;        // time index of this step
    const int tp = d == 0 ? tt - 1 : tt + 1;                // time index of the previous step of this direction
    const bool first = a.step == 0;
    const int i15 = lane & 15, g4 = lane >> 4;

    // operands of this thread's cell updates ((b, u) pairs p = tid + 256 i), fetched before the matrix part so their
    // latency overlaps it
    float gxr[NRT][4], cpr[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int p = tid + 256 * i, b = p >> 4, u = p & 15;
        const bool ok = b < a.B;
        const float* gx = a.gx + ((int64_t)(ok ? b : 0) * S + tt) * a.ldg + (int64_t)d * 4 * H + u0 + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) gxr[i][q] = gx[q * H];
        cpr[i] = first ? 0.f : a.c_all[(((int64_t)(ok ? b : 0) * S + tp) * 2 + d) * H + u0 + u];
    }
    f32x4 acc[NRT];
#pragma unroll
    for (int r = 0; r < NRT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) {
        // wave = gate: rows (wave*H + u0 + i15) of W_hh[d], against h_{t-1} rows b
        const bf16_t* wrow = a.whh + ((int64_t)d * 4 * H + (int64_t)wave * H + u0 + i15) * H + 8 * g4;
        const bf16_t* hbase = a.y + (int64_t)tp * 2 * H + (int64_t)d * H + 8 * g4;
        // UB k-steps (all 24 of H = 768 for B <= 32) of independent 16-byte loads are issued before their MFMAs: the
        // operands come from L2 / Infinity Cache and every dependent round trip costs ~2 us of a ~10 us step
        for (int kb = 0; kb < H; kb += 32 * UB) {
            bf16x8 wf[UB], hf[NRT][UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k0 = kb + 32 * u;
                const bool kok = k0 < H;
                wf[u] = frag16(wrow + k0, kok);
#pragma unroll
                for (int r = 0; r < NRT; ++r) {
                        const int b = 16 * r + i15;
                        hf[r][u] = frag16(hbase + (int64_t)b * S * 2 * H + k0, kok && b < a.B);
                    }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r] = mfma16(wf[u], hf[r][u], acc[r]);   // D[i = unit 4*g4 + q][j = batch i15]
        }
    }
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_g[wave][16 * r + i15][4 * g4 + q] = acc[r][q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int p = tid + 256 * i, b = p >> 4, u = p & 15;
        if (b >= a.B) continue;
        const int64_t row = (int64_t)b * S + tt;
        const float gi = sigmoid_f(s_g[0][b][u] + gxr[i][0]);
        const float gf = sigmoid_f(s_g[1][b][u] + gxr[i][1]);
        const float gg = tanhf(s_g[2][b][u] + gxr[i][2]);
        const float go = sigmoid_f(s_g[3][b][u] + gxr[i][3]);
        const float cp = cpr[i];
        const float c = gf * cp + gi * gg;
        const float h = go * tanhf(c);
        a.c_all[(row * 2 + d) * H + u0 + u] = c;
        a.y[row * 2 * H + (int64_t)d * H + u0 + u] = f2bf(h);
        bf16_t* act = a.act + (row * 2 + d) * 4 * H + u0 + u;
        act[0] = f2bf(gi); act[H] = f2bf(gf); act[2 * H] = f2bf(gg); act[3 * H] = f2bf(go);
        if (a.hprev)
            a.hprev[row * 2 * H + (int64_t)d * H + u0 + u] =
                first ? f2bf(0.f) : a.y[((int64_t)b * S + tp) * 2 * H + (int64_t)d * H + u0 + u];
    }
}

// --------------------------------------------------------------------------------------------------- backward step
// a.step counts the forward steps; the launches run step = S-1 .. 0.
template <int NRT, int UB>
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const LstmArgs a) {
    __shared__ float s_p[4][MAX_RT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, u0 = blockIdx.x * 16, H = a.H, S = a.S;
    const int tt = d == 0 ? a.step : S - 1 - a.step;
    const int tp = d == 0 ? tt - 1 : tt + 1;                // previous forward step
    const int tn = d == 0 ? tt + 1 : tt - 1;                // next forward step (its dgates are already computed)
    const bool last = a.step == S - 1, first = a.step == 0;
    const int i15 = lane & 15, g4 = lane >> 4;

    // this thread's elementwise operands, fetched before the matrix part
    float dyr[NRT], actr[NRT][4], cr[NRT], cpr[NRT], carr[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int p = tid + 256 * i, b = p >> 4, u = p & 15;
        const int bb = b < a.B ? b : 0;
        const int64_t row = (int64_t)bb * S + tt;
        dyr[i] = bf2f(a.dy[row * 2 * H + (int64_t)d * H + u0 + u]);
        const bf16_t* act = a.act + (row * 2 + d) * 4 * H + u0 + u;
#pragma unroll
        for (int q = 0; q < 4; ++q) actr[i][q] = bf2f(act[q * H]);
        cr[i] = a.c_all[(row * 2 + d) * H + u0 + u];
        cpr[i] = first ? 0.f : a.c_all[(((int64_t)bb * S + tp) * 2 + d) * H + u0 + u];
        carr[i] = last ? 0.f : a.dc_carry[((int64_t)d * a.B + bb) * H + u0 + u];
    }
    f32x4 acc[NRT];
#pragma unroll
    for (int r = 0; r < NRT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!last) {
        // dh_rec[b][u] = sum_k dgates_next[b][k] * W_hh[k][u], k over 4H: wave w takes k in [w*H, (w+1)*H)
        const bf16_t* wrow = a.whh + ((int64_t)d * H + u0 + i15) * 4 * H + (int64_t)wave * H + 8 * g4;   // W_hh^T rows
        const bf16_t* gbase = a.dgates + (int64_t)tn * a.ldg + (int64_t)d * 4 * H + (int64_t)wave * H + 8 * g4;
        for (int kb = 0; kb < H; kb += 32 * UB) {
            bf16x8 wf[UB], gf[NRT][UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k0 = kb + 32 * u;
                const bool kok = k0 < H;
                wf[u] = frag16(wrow + k0, kok);
#pragma unroll
                for (int r = 0; r < NRT; ++r) {
                        const int b = 16 * r + i15;
                        gf[r][u] = frag16(gbase + (int64_t)b * S * a.ldg + k0, kok && b < a.B);
                    }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r] = mfma16(wf[u], gf[r][u], acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_p[wave][16 * r + i15][4 * g4 + q] = acc[r][q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int p = tid + 256 * i, b = p >> 4, u = p & 15;
        if (b >= a.B) continue;
        const int64_t row = (int64_t)b * S + tt;
        const float dh = dyr[i] + (s_p[0][b][u] + s_p[1][b][u] + s_p[2][b][u] + s_p[3][b][u]);
        const float gi = actr[i][0], gf = actr[i][1], gg = actr[i][2], go = actr[i][3];
        const float c = cr[i], cp = cpr[i];
        const float tc = tanhf(c);
        const float dc = dh * go * (1.f - tc * tc) + carr[i];
        a.dc_carry[((int64_t)d * a.B + b) * H + u0 + u] = dc * gf;
        bf16_t* dg = a.dgates + row * a.ldg + (int64_t)d * 4 * H + u0 + u;
        dg[0] = f2bf(dc * gg * gi * (1.f - gi));
        dg[H] = f2bf(dc * cp * gf * (1.f - gf));
        dg[2 * H] = f2bf(dc * gi * (1.f - gg * gg));
        dg[3 * H] = f2bf(dh * tc * go * (1.f - go));
    }
}

// =====================================================================================================================
// Persistent recurrence: ONE launch walks all S steps.  Same decomposition (a block per 16 hidden units and direction,
// a wave per gate), but the block keeps its W_hh slice (4 gates x 16 units x H, KS fragments per wave) in registers
// for the whole sequence and its cell state in registers; per step only h_{t-1} (B x H bf16, written by the other
// blocks of the direction) crosses HBM/L2.  Steps are separated by a per-direction grid barrier: every block
// publishes its h slice (device-scope release fence + atomic ticket) and spins (with s_sleep) until all H/16 blocks of
// its direction have published step t.  All blocks are co-resident by construction (H/16 * 2 <= 256 blocks, one per
// CU: checked on the host), so the barrier cannot deadlock; the spin is bounded anyway (it gives up after ~2^22 polls,
// raises the error word and finishes with wrong data rather than hanging the device).
struct LstmPersist {
    unsigned int* tickets;   // [2] per-direction step counters, zeroed by the host before the launch
    unsigned int* err;       // [1] set to 1 if a barrier wait gave up
};

// Hand-off form (cdna_hip_programming.md Guideline 16): the payload another block reads (h_t / dgates_t) is stored
// WRITE-THROUGH with 8-byte agent-scope atomic stores (sc1), every storing wave drains its stores (s_waitcnt vmcnt(0)),
// the block barrier joins them and ONE lane adds the ticket; the consumer polls with one lane, that lane issues ONE
// agent-scope acquire (invalidates this CU's L1) and drains it, the block barrier releases the other waves to plain
// loads.  No release fence (buffer_wbl2) and no per-thread __threadfence(): those made a step ~26 us.
typedef __attribute__((address_space(1))) unsigned long long gu64_t;
__device__ __forceinline__ void lstm_store_wt(void* p, unsigned long long v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A wait that gives up (a peer block that is not resident, a lost word) must not pass for a result: it raises the error
// word -- host-visible memory when the library could map it (lstm_err_word), so the host reads it without a device
// synchronisation -- and POISONS the recurrence: the waiting block continues with NaN operands, which reach every later
// step of every block through h / dgates, so y, the loss and the gradients of the call are NaN.  Once poisoned a block no
// longer spins (a failed launch costs one time-out, not one per step).
__device__ __forceinline__ void lstm_raise(unsigned int* err) {
    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ float lstm_nan() { return __uint_as_float(0x7fc00000u); }
// returns true (to every thread of the block) when the block is poisoned; ``s_poison`` is a __shared__ word zeroed at kernel start
__device__ __forceinline__ bool lstm_grid_wait(unsigned int* ctr, unsigned int target, unsigned int* err, int limit,
                                               int* s_poison) {
    if (threadIdx.x == 0) {
        int polls = 0;
        if (*s_poison == 0)
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++polls > limit) { lstm_raise(err); *s_poison = 1; break; }
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return *s_poison != 0;
}
__device__ __forceinline__ void lstm_grid_signal(unsigned int* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its write-through stores
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NRT, int KS>
__global__ __launch_bounds__(256) void lstm_fwd_persistent_kernel(const LstmArgs a, const LstmPersist ps) {
    __shared__ float s_g[4][NRT * 16][17];
    __shared__ __attribute__((aligned(16))) bf16_t s_h[NRT * 16][16];
    __shared__ int s_poison;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, u0 = blockIdx.x * 16, H = a.H, S = a.S;
    const int i15 = lane & 15, g4 = lane >> 4;
    const unsigned int nblk = gridDim.x;
    if (tid == 0) s_poison = 0;
    __syncthreads();
    bf16x8 wf[KS];   // this wave's gate rows of W_hh, resident for the whole sequence
    {
        const bf16_t* wrow = a.whh + ((int64_t)d * 4 * H + (int64_t)wave * H + u0 + i15) * H + 8 * g4;
#pragma unroll
        for (int u = 0; u < KS; ++u) wf[u] = frag16(wrow + 32 * u, 32 * u < H);
    }
    float creg[NRT];   // cell state of this thread's (b, u) pairs
#pragma unroll
    for (int i = 0; i < NRT; ++i) creg[i] = 0.f;
    for (int step = 0; step < S; ++step) {
        const int tt = d == 0 ? step : S - 1 - step;
        const int tp = d == 0 ? tt - 1 : tt + 1;
        float gxr[NRT][4];
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            const float* gx = a.gx + ((int64_t)(b < a.B ? b : 0) * S + tt) * a.ldg + (int64_t)d * 4 * H + u0 + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) gxr[i][q] = gx[q * H];
        }
        f32x4 acc[NRT];
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            const bool bad = lstm_grid_wait(ps.tickets + d, (unsigned)step * nblk, ps.err, a.poll_limit, &s_poison);
            const bf16_t* hbase = a.y + (int64_t)tp * 2 * H + (int64_t)d * H + 8 * g4;
            bf16x8 hf[NRT][KS];
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r) {
                    const int b = 16 * r + i15;
                    hf[r][u] = frag16(hbase + (int64_t)b * S * 2 * H + 32 * u, 32 * u < H && b < a.B);
                }
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r] = mfma16(wf[u], hf[r][u], acc[r]);
            if (bad) {
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r] = f32x4{lstm_nan(), lstm_nan(), lstm_nan(), lstm_nan()};
            }
        }
        __syncthreads();   // s_g of the previous step fully consumed
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_g[wave][16 * r + i15][4 * g4 + q] = acc[r][q];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            if (b >= a.B) continue;
            const int64_t row = (int64_t)b * S + tt;
            const float gi = sigmoid_f(s_g[0][b][u] + gxr[i][0]);
            const float gf = sigmoid_f(s_g[1][b][u] + gxr[i][1]);
            const float gg = tanhf(s_g[2][b][u] + gxr[i][2]);
            const float go = sigmoid_f(s_g[3][b][u] + gxr[i][3]);
            const float c = gf * creg[i] + gi * gg;
            const float h = go * tanhf(c);
            if (a.hprev)   // h_{t-1} of this (b, u): still in s_h from the previous step (same thread wrote it)
                a.hprev[row * 2 * H + (int64_t)d * H + u0 + u] = step == 0 ? f2bf(0.f) : s_h[b][u];
            creg[i] = c;
            a.c_all[(row * 2 + d) * H + u0 + u] = c;
            s_h[b][u] = f2bf(h);
            bf16_t* act = a.act + (row * 2 + d) * 4 * H + u0 + u;
            act[0] = f2bf(gi); act[H] = f2bf(gf); act[2 * H] = f2bf(gg); act[3 * H] = f2bf(go);
        }
        __syncthreads();
        // publish h_t: 4 units (8 bytes) per store, write-through
        for (int p = tid; p < NRT * 16 * 4; p += 256) {
            const int b = p >> 2, part = p & 3;
            if (b < a.B)
                lstm_store_wt(a.y + ((int64_t)b * S + tt) * 2 * H + (int64_t)d * H + u0 + 4 * part,
                              *reinterpret_cast<const unsigned long long*>(&s_h[b][4 * part]));
        }
        // (a block that gave up publishes its NaN slice but takes no more tickets: tickets of a block that runs ahead without
        //  waiting would lift the count over the target of blocks still waiting and release them with finite, stale operands)
        if (step + 1 < S && s_poison == 0 && !(a.test_drop == step && blockIdx.x == 0 && blockIdx.y == 0)) lstm_grid_signal(ps.tickets + d);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Flag-in-data hand-off ("LL" form): h_t is published as 8-byte words {2 hidden units bf16, u32 step tag}, written with
// ONE agent-scope 8-byte store each, and the consumers poll the data words themselves until every tag reads `step`.
// Against the ticket form above this removes, per time step, the producer's store drain + ticket add and the consumer's
// separate poll + L1 invalidate: one store -> load hop instead of four dependent round trips.  Words alternate between two
// buffers by step parity: a block can publish step t+1 only after it has read every block's step t, and nobody overwrites
// parity p before all of step t+1 (which needs every block's reads of step t done) has been published.  The buffer is
// zeroed by the host before each launch, tags are step + 1.
// The reduction over H is split over the 4 waves (wave w: columns [w H/4, (w+1) H/4) of W_hh for ALL four gates), so a
// wave polls only its quarter of h_t (24 KB of words at B = 32, H = 768) and the block reads h once, not once per gate.
constexpr int LSTM_LL_MAXH = 1024, LSTM_LL_ROWS = 16 * MAX_RT;
__device__ unsigned long long g_lstm_ll[2 * 2 * LSTM_LL_ROWS * (LSTM_LL_MAXH / 2)];   // [parity][dir][b][H / 2]

__device__ __forceinline__ u32x4 ll_load16(const unsigned long long* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int NRT, int KQ>   // KQ = W_hh fragments per gate and wave = H / 128
__global__ __launch_bounds__(256) void lstm_fwd_ll_kernel(const LstmArgs a, unsigned int* err) {
    __shared__ float s_g[4][4][NRT * 16][17];   // [wave][gate][b][unit]
    __shared__ __attribute__((aligned(16))) bf16_t s_h[NRT * 16][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, u0 = blockIdx.x * 16, H = a.H, S = a.S;
    const int i15 = lane & 15, g4 = lane >> 4;
    const int kq0 = wave * (H >> 2);
    const int bofs = 16 * blockIdx.z;   // batch rows are independent recurrences: grid.z may split them into tiles of 16
    bf16x8 wf[4][KQ];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bf16_t* wrow = a.whh + ((int64_t)d * 4 * H + (int64_t)q * H + u0 + i15) * H + kq0 + 8 * g4;
#pragma unroll
        for (int u = 0; u < KQ; ++u) wf[q][u] = frag16(wrow + 32 * u, true);
    }
    float creg[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) creg[i] = 0.f;
    bool poison = false;   // wave-uniform: a wait of this wave gave up (lstm_raise); NaN operands from then on, no more spinning
    for (int step = 0; step < S; ++step) {
        const int tt = d == 0 ? step : S - 1 - step;
        float gxr[NRT][4];
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            const float* gx = a.gx + ((int64_t)(b + bofs < a.B ? b + bofs : 0) * S + tt) * a.ldg + (int64_t)d * 4 * H + u0 + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) gxr[i][q] = gx[q * H];
        }
        f32x4 acc[NRT][4];
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step > 0) {
            // h_{t-1}: words of step - 1 (tag = step) in buffer (step - 1) & 1
            const unsigned long long* base = g_lstm_ll + ((int64_t)(((step - 1) & 1) * 2 + d) * LSTM_LL_ROWS) * (LSTM_LL_MAXH / 2) +
                                             ((kq0 + 8 * g4) >> 1);
            const uint32_t want = (uint32_t)step;
            bf16x8 hf[NRT][KQ];
            int polls = poison ? a.poll_limit : 0;
            if (!poison) {   // cheap poll first: the last word (row B-1, units 14..15) of each producer block of this wave's quarter -- a
                // hint only (words become visible in any order); the full load below checks every tag
                const int nsrc = H >> 6;   // (H / 4) / 16 producer blocks
                const int blast = (a.B < bofs + 16 * NRT ? a.B : bofs + 16 * NRT) - 1;
                const unsigned long long* sp = base - ((8 * g4) >> 1) + (int64_t)blast * (LSTM_LL_MAXH / 2) + 8 * (lane < nsrc ? lane : 0) + 7;
                while (true) {
                    const unsigned long long v = __hip_atomic_load(sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64((uint32_t)(v >> 32) != want) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > a.poll_limit) break;   // (the full check below reports it)
                }
            }
            bool pending = true;
            while (pending) {
                u32x4 raw[NRT][KQ][2];
#pragma unroll
                for (int r = 0; r < NRT; ++r)
#pragma unroll
                    for (int u = 0; u < KQ; ++u) {
                        const unsigned long long* p = base + (int64_t)(bofs + 16 * r + i15) * (LSTM_LL_MAXH / 2) + 16 * u;
                        raw[r][u][0] = ll_load16(p);
                        raw[r][u][1] = ll_load16(p + 2);
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (the loads are asm outputs the compiler knows nothing about: tie every register to a volatile asm that
                //  follows the wait, so that no use is scheduled ahead of it)
#pragma unroll
                for (int r = 0; r < NRT; ++r)
#pragma unroll
                    for (int u = 0; u < KQ; ++u) {
                        asm volatile("" : "+v"(raw[r][u][0]));
                        asm volatile("" : "+v"(raw[r][u][1]));
                    }
                bool ok = true;
#pragma unroll
                for (int r = 0; r < NRT; ++r) {
                    const bool live = bofs + 16 * r + i15 < a.B;
#pragma unroll
                    for (int u = 0; u < KQ; ++u) {
                        ok = ok && (!live || (raw[r][u][0][1] == want && raw[r][u][0][3] == want && raw[r][u][1][1] == want &&
                                              raw[r][u][1][3] == want));
                        const u32x4 dw = {raw[r][u][0][0], raw[r][u][0][2], raw[r][u][1][0], raw[r][u][1][2]};
                        hf[r][u] = live ? as_bf16x8(dw) : as_bf16x8(u32x4{0u, 0u, 0u, 0u});
                    }
                }
                pending = __builtin_amdgcn_ballot_w64(!ok) != 0ull;
                if (pending) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > a.poll_limit) {
                        if (lane == 0) lstm_raise(err);
                        pending = false;
                        poison = true;
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < KQ; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[r][q] = mfma16(wf[q][u], hf[r][u], acc[r][q]);
            if (poison) {
#pragma unroll
                for (int r = 0; r < NRT; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[r][q] = f32x4{lstm_nan(), lstm_nan(), lstm_nan(), lstm_nan()};
            }
        }
        __syncthreads();   // s_g / s_h of the previous step fully consumed
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) s_g[wave][q][16 * r + i15][4 * g4 + e] = acc[r][q][e];
        __syncthreads();
        float gi_[NRT], gf_[NRT], gg_[NRT], go_[NRT], hp_[NRT];
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) z[q] = gxr[i][q] + ((s_g[0][q][b][u] + s_g[1][q][b][u]) + (s_g[2][q][b][u] + s_g[3][q][b][u]));
            gi_[i] = sigmoid_f(z[0]); gf_[i] = sigmoid_f(z[1]); gg_[i] = tanhf(z[2]); go_[i] = sigmoid_f(z[3]);
            const float c = gf_[i] * creg[i] + gi_[i] * gg_[i];
            hp_[i] = step == 0 ? 0.f : bf2f(s_h[b][u]);   // h_{t-1} of this (b, u): the same thread wrote it last step
            creg[i] = c;
            s_h[b][u] = f2bf(go_[i] * tanhf(c));
        }
        __syncthreads();
        // publish h_t first: 2 units + tag per 8-byte word, write-through
        if (step + 1 < S && !(a.test_drop == step && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)) {
            unsigned long long* out = g_lstm_ll + ((int64_t)((step & 1) * 2 + d) * LSTM_LL_ROWS) * (LSTM_LL_MAXH / 2) + (u0 >> 1);
            for (int p = tid; p < NRT * 16 * 8; p += 256) {
                const int b = p >> 3, j = p & 7;
                if (b + bofs < a.B) {
                    const uint32_t dw = *reinterpret_cast<const uint32_t*>(&s_h[b][2 * j]);
                    __hip_atomic_store(out + (int64_t)(b + bofs) * (LSTM_LL_MAXH / 2) + j, (unsigned long long)dw | ((unsigned long long)(step + 1) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // ... then everything nobody waits for: y, the saved cell state / activations / h_{t-1}
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            if (b + bofs >= a.B) continue;
            const int64_t row = (int64_t)(b + bofs) * S + tt;
            a.y[row * 2 * H + (int64_t)d * H + u0 + u] = s_h[b][u];
            if (a.hprev) a.hprev[row * 2 * H + (int64_t)d * H + u0 + u] = f2bf(hp_[i]);
            a.c_all[(row * 2 + d) * H + u0 + u] = creg[i];
            bf16_t* act = a.act + (row * 2 + d) * 4 * H + u0 + u;
            act[0] = f2bf(gi_[i]); act[H] = f2bf(gf_[i]); act[2 * H] = f2bf(gg_[i]); act[3 * H] = f2bf(go_[i]);
        }
    }
}

template <int NRT, int KS>
__global__ __launch_bounds__(256) void lstm_bwd_persistent_kernel(const LstmArgs a, const LstmPersist ps) {
    __shared__ float s_p[4][NRT * 16][17];
    __shared__ __attribute__((aligned(16))) bf16_t s_dg[4][NRT * 16][16];
    __shared__ int s_poison;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, u0 = blockIdx.x * 16, H = a.H, S = a.S;
    const int i15 = lane & 15, g4 = lane >> 4;
    const unsigned int nblk = gridDim.x;
    if (tid == 0) s_poison = 0;
    __syncthreads();
    bf16x8 wf[KS];   // rows u0.. of W_hh^T, this wave's quarter of the 4H reduction
    {
        const bf16_t* wrow = a.whh + ((int64_t)d * H + u0 + i15) * 4 * H + (int64_t)wave * H + 8 * g4;
#pragma unroll
        for (int u = 0; u < KS; ++u) wf[u] = frag16(wrow + 32 * u, 32 * u < H);
    }
    float carry[NRT];   // dc_{t+1} * f_{t+1} of this thread's (b, u) pairs
#pragma unroll
    for (int i = 0; i < NRT; ++i) carry[i] = 0.f;
    for (int step = S - 1; step >= 0; --step) {
        const int tt = d == 0 ? step : S - 1 - step;
        const int tp = d == 0 ? tt - 1 : tt + 1;
        const int tn = d == 0 ? tt + 1 : tt - 1;
        const bool first = step == 0;
        float dyr[NRT], actr[NRT][4], cr[NRT], cpr[NRT];
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            const int bb = b < a.B ? b : 0;
            const int64_t row = (int64_t)bb * S + tt;
            dyr[i] = bf2f(a.dy[row * 2 * H + (int64_t)d * H + u0 + u]);
            const bf16_t* act = a.act + (row * 2 + d) * 4 * H + u0 + u;
#pragma unroll
            for (int q = 0; q < 4; ++q) actr[i][q] = bf2f(act[q * H]);
            cr[i] = a.c_all[(row * 2 + d) * H + u0 + u];
            cpr[i] = first ? 0.f : a.c_all[(((int64_t)bb * S + tp) * 2 + d) * H + u0 + u];
        }
        f32x4 acc[NRT];
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (step < S - 1) {
            const bool bad = lstm_grid_wait(ps.tickets + d, (unsigned)(S - 1 - step) * nblk, ps.err, a.poll_limit, &s_poison);
            const bf16_t* gbase = a.dgates + (int64_t)tn * a.ldg + (int64_t)d * 4 * H + (int64_t)wave * H + 8 * g4;
            bf16x8 gf[NRT][KS];
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r) {
                    const int b = 16 * r + i15;
                    gf[r][u] = frag16(gbase + (int64_t)b * S * a.ldg + 32 * u, 32 * u < H && b < a.B);
                }
#pragma unroll
            for (int u = 0; u < KS; ++u)
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r] = mfma16(wf[u], gf[r][u], acc[r]);
            if (bad) {
#pragma unroll
                for (int r = 0; r < NRT; ++r) acc[r] = f32x4{lstm_nan(), lstm_nan(), lstm_nan(), lstm_nan()};
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) s_p[wave][16 * r + i15][4 * g4 + q] = acc[r][q];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const int p = tid + 256 * i, b = p >> 4, u = p & 15;
            if (b >= a.B) continue;
            (void)tt;
            const float dh = dyr[i] + (s_p[0][b][u] + s_p[1][b][u] + s_p[2][b][u] + s_p[3][b][u]);
            const float gi = actr[i][0], gf = actr[i][1], gg = actr[i][2], go = actr[i][3];
            const float tc = tanhf(cr[i]);
            const float dc = dh * go * (1.f - tc * tc) + carry[i];
            carry[i] = dc * gf;
            s_dg[0][b][u] = f2bf(dc * gg * gi * (1.f - gi));
            s_dg[1][b][u] = f2bf(dc * cpr[i] * gf * (1.f - gf));
            s_dg[2][b][u] = f2bf(dc * gi * (1.f - gg * gg));
            s_dg[3][b][u] = f2bf(dh * tc * go * (1.f - go));
        }
        __syncthreads();
        // publish dgates_t: 4 units (8 bytes) per store, write-through
        for (int p = tid; p < 4 * NRT * 16 * 4; p += 256) {
            const int gate = p / (NRT * 64), rem = p % (NRT * 64), b = rem >> 2, part = rem & 3;
            if (b < a.B)
                lstm_store_wt(a.dgates + ((int64_t)b * S + tt) * a.ldg + (int64_t)d * 4 * H + (int64_t)gate * H + u0 +
                                  4 * part,
                              *reinterpret_cast<const unsigned long long*>(&s_dg[gate][b][4 * part]));
        }
        if (step > 0 && s_poison == 0 && !(a.test_drop == step && blockIdx.x == 0 && blockIdx.y == 0)) lstm_grid_signal(ps.tickets + d);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward recurrence, reduce-scatter form with the flag-in-data hand-off.  dh_t = dy_t + dgates_{t+1} . W_hh needs ALL 4H
// gate columns of step t+1 in every block if the blocks exchange dgates (the ticket kernel above: 192 KB per block and step,
// and as tagged words twice that through L2-bypassing loads -- measured 14.6 vs 8.9 us per step).  Here the PRODUCER
// multiplies: the block that owns units U computes its own 64 gate columns of dgates_t and at once their contribution
//     P = dgates_t[:, G(U)] . W_hh[G(U), :]      (16 rows x H, K = 64: 2 MFMA k-steps per 16-column tile)
// to dh_{t-1} of EVERY unit, and sends each 16-column tile of P to the block that owns those units as {2 x bf16, tag} words;
// a consumer sums the H / 16 partial tiles addressed to it (12 x 16-byte loads per thread at H = 768).  Every word is written
// once and read once (no broadcast), batch tiles of 16 rows are independent blocks, and a step is one store -> load hop.
// Word buffer: [parity][dir][dest block][src block][32 rows][8 words]; tags S - step, parity by step (same argument as the
// forward kernel: a block reaches step t-1 only after every block has produced step t, i.e. consumed step t+1).
template <int NT>   // 16-column tiles of P per wave = sources summed per wave = H / 64
__global__ __launch_bounds__(256) void lstm_bwd_rs_kernel(const LstmArgs a, unsigned long long* llr, unsigned int* err) {
    __shared__ float s_p[4][16][17];
    __shared__ __attribute__((aligned(16))) bf16_t s_dg[4][16][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = blockIdx.y, blk = blockIdx.x, u0 = blk * 16, H = a.H, S = a.S;
    const int nblk = gridDim.x, bofs = 16 * blockIdx.z;
    const int i15 = lane & 15, g4 = lane >> 4;
    // W_hh^T rows n (all H of them, this wave's NT tiles), columns = this block's 64 gate columns in the order k = 16 q + uu
    bf16x8 wf[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = 16 * (wave * NT + j) + i15;
#pragma unroll
        for (int f = 0; f < 2; ++f)
            wf[j][f] = frag16(a.whh + ((int64_t)d * H + n) * 4 * H + (int64_t)(2 * f + (g4 >> 1)) * H + u0 + 8 * (g4 & 1), true);
    }
    const int64_t region = (int64_t)nblk * nblk * 256;   // words per (parity, direction)
    const int cb = tid >> 4, cu = tid & 15;               // this thread's (row, unit) of the cell update
    const bool crow = bofs + cb < a.B;
    const int br = lane >> 2, wp = lane & 3;              // this lane's (row, 4-unit group) of the partial tiles
    const bool lrow = bofs + br < a.B;
    float carry = 0.f;
    bool poison = false;   // wave-uniform, as in lstm_fwd_ll_kernel
    for (int step = S - 1; step >= 0; --step) {
        const int tt = d == 0 ? step : S - 1 - step;
        const int tp = d == 0 ? tt - 1 : tt + 1;
        const bool first = step == 0;
        const int64_t row = (int64_t)(crow ? bofs + cb : 0) * S + tt;
        const float dyr = bf2f(a.dy[row * 2 * H + (int64_t)d * H + u0 + cu]);
        float actr[4];
        {
            const bf16_t* act = a.act + (row * 2 + d) * 4 * H + u0 + cu;
#pragma unroll
            for (int q = 0; q < 4; ++q) actr[q] = bf2f(act[q * H]);
        }
        const float cr = a.c_all[(row * 2 + d) * H + u0 + cu];
        const float cpr = first ? 0.f : a.c_all[(((int64_t)(crow ? bofs + cb : 0) * S + tp) * 2 + d) * H + u0 + cu];
        float part[4] = {0.f, 0.f, 0.f, 0.f};
        if (step < S - 1) {
            // partial tiles of step + 1 addressed to this block: tag S - 1 - step, buffer (step + 1) & 1; wave w sums the
            // sources w NT .. w NT + NT - 1
            const uint32_t want = (uint32_t)(S - 1 - step);
            const unsigned long long* base = llr + ((int64_t)(((step + 1) & 1) * 2 + d)) * region +
                                             (((int64_t)blk * nblk + wave * NT) * 32 + bofs + br) * 8 + 2 * wp;
            int polls = poison ? a.poll_limit : 0;
            bool pending = true;
            while (pending) {
                u32x4 raw[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) raw[k] = ll_load16(base + (int64_t)k * 256);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < NT; ++k) asm volatile("" : "+v"(raw[k]));
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 4; ++e) part[e] = 0.f;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    ok = ok && (!lrow || (raw[k][1] == want && raw[k][3] == want));
                    part[0] += __uint_as_float(raw[k][0] << 16);
                    part[1] += __uint_as_float(raw[k][0] & 0xffff0000u);
                    part[2] += __uint_as_float(raw[k][2] << 16);
                    part[3] += __uint_as_float(raw[k][2] & 0xffff0000u);
                }
                pending = __builtin_amdgcn_ballot_w64(!ok) != 0ull;
                if (pending) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > a.poll_limit) {   // (default 2^17 ~ 0.2 s: a lost word.  Report + poison instead of hanging the device.)
                        if (lane == 0) lstm_raise(err);
                        pending = false;
                        poison = true;
                    }
                }
            }
            if (!lrow) { part[0] = part[1] = part[2] = part[3] = 0.f; }
            if (poison) { part[0] = part[1] = part[2] = part[3] = lstm_nan(); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s_p[wave][br][4 * wp + e] = part[e];
        __syncthreads();
        if (crow) {
            const float dh = dyr + ((s_p[0][cb][cu] + s_p[1][cb][cu]) + (s_p[2][cb][cu] + s_p[3][cb][cu]));
            const float gi = actr[0], gf = actr[1], gg = actr[2], go = actr[3];
            const float tc = tanhf(cr);
            const float dc = dh * go * (1.f - tc * tc) + carry;
            carry = dc * gf;
            s_dg[0][cb][cu] = f2bf(dc * gg * gi * (1.f - gi));
            s_dg[1][cb][cu] = f2bf(dc * cpr * gf * (1.f - gf));
            s_dg[2][cb][cu] = f2bf(dc * gi * (1.f - gg * gg));
            s_dg[3][cb][cu] = f2bf(dh * tc * go * (1.f - go));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) s_dg[q][cb][cu] = f2bf(0.f);
        }
        __syncthreads();
        if (step > 0 && !(a.test_drop == step && blk == 0 && d == 0 && blockIdx.z == 0)) {
            // P tiles of this step: 2 MFMA k-steps each, sent as tagged words to the owners of the columns
            const bf16x8 g0 = as_bf16x8(*reinterpret_cast<const u32x4*>(&s_dg[g4 >> 1][i15][8 * (g4 & 1)]));
            const bf16x8 g1 = as_bf16x8(*reinterpret_cast<const u32x4*>(&s_dg[2 + (g4 >> 1)][i15][8 * (g4 & 1)]));
            unsigned long long* out = llr + ((int64_t)((step & 1) * 2 + d)) * region +
                                      (((int64_t)(wave * NT) * nblk + blk) * 32 + bofs + i15) * 8 + 2 * g4;
            const uint32_t tagw = (uint32_t)(S - step);
            const bool live = bofs + i15 < a.B;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                acc = mfma16(wf[j][0], g0, acc);
                acc = mfma16(wf[j][1], g1, acc);
                if (live) {
                    // two tagged words in ONE 16-byte write-through store (each 8-byte word validates itself, so it does
                    // not matter whether the 16 bytes become visible together)
                    const u32x4 w = {pack_bf16x2(acc[0], acc[1]), tagw, pack_bf16x2(acc[2], acc[3]), tagw};
                    unsigned long long* o = out + (int64_t)j * nblk * 256;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(o), "v"(w) : "memory");
                }
            }
        }
        // the plain copy of dgates_t (operand of the weight / input gradient GEMMs after the loop): one 8-byte piece per thread
        {
            const int gate = tid >> 6, rem = tid & 63, b = rem >> 2, partq = rem & 3;
            if (bofs + b < a.B)
                *reinterpret_cast<unsigned long long*>(a.dgates + ((int64_t)(bofs + b) * S + tt) * a.ldg + (int64_t)d * 4 * H +
                                                       (int64_t)gate * H + u0 + 4 * partq) =
                    *reinterpret_cast<const unsigned long long*>(&s_dg[gate][b][4 * partq]);
        }
    }
}

// The persistent launches wait on each other's blocks: every block of the grid must be resident at once, i.e. the grid may
// not exceed the CUs of THIS device (256 on a whole MI355X, fewer in a partitioned mode); larger grids take one launch per step.
// icka_lstm_set_reserved_cus: CUs kept free of persistent LSTM blocks.  A block of these kernels holds 352 of a SIMD's 512
// registers per lane (one wave per SIMD), so a CU on which another kernel's workgroup already occupies more than 160
// registers per lane of some SIMD -- an RCCL all-reduce workgroup on the communication stream of a data-parallel step
// (dp.GradReducer) -- cannot take one until that workgroup exits: the grid is no longer co-resident by construction and
// its blocks would spin on a peer that has not started.  The reducer reserves as many CUs as RCCL runs channels.
static int lstm_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else { n = 1; (void)hipGetLastError(); }
    }
    const int avail = n - g_icka_reserved_cus;
    return avail > 0 ? avail : 0;
}
// Which form a call takes is decided PER CALL by its ``flags`` argument (ICKA_LSTM_*, include/icka_hip.h): the default (0) is
// the persistent launch with the flag-in-data hand-off and batch tiles of 16 rows as separate blocks; there is no process-wide
// switch (tests and tools/lstm_bench.py pass the flags of the form they want).
static unsigned long long* lstm_rs_words(hipStream_t st) {   // [2 parity][2 dir][64 dest][64 src][32 rows][8 words]: 33.5 MB, allocated once
    static unsigned long long* p = nullptr;
    static bool failed = false;
    if (!p && !failed) {
        // never allocate while the stream is being captured into a graph (hipMalloc would invalidate the capture): the
        // caller then uses the ticket form for this launch; the first eager launch allocates
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
        if (hipMalloc((void**)&p, sizeof(unsigned long long) * 4 * 64 * 64 * 256) != hipSuccess) { p = nullptr; failed = true; (void)hipGetLastError(); }
    }
    return p;
}
static unsigned long long* lstm_ll_words() {
    static unsigned long long* p = nullptr;
    if (!p && hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_lstm_ll)) != hipSuccess) p = nullptr;
    return p;
}
__device__ unsigned int g_lstm_sync[8];   // [0..1] tickets per direction (2..3 spare), [4] error word (fallback, see lstm_err_word)
static unsigned int* lstm_sync_words() {   // address looked up once (not a stream operation: safe under graph capture)
    static unsigned int* p = nullptr;
    if (!p && hipGetSymbolAddress((void**)&p, HIP_SYMBOL(g_lstm_sync)) != hipSuccess) p = nullptr;
    return p;
}
// The error word of the persistent recurrences lives in HOST memory mapped into the device (a kernel raises it with one
// system-scope store, lstm_raise; the host reads it as plain memory: icka_lstm_barrier_error costs no device
// synchronisation and may be polled at every host touch-point).  Allocated on the first call that is not inside a stream
// capture; until then (and if the mapping fails) the device word g_lstm_sync[4] is used and read back with a memcpy.
static volatile unsigned int* g_lstm_err_host = nullptr;
static unsigned int* g_lstm_err_dev = nullptr;
static bool g_lstm_fallback_used = false;   // a launch was given the device word g_lstm_sync[4] as its error word
static unsigned int* lstm_err_word(hipStream_t st) {
    if (g_lstm_err_dev) return g_lstm_err_dev;
    static bool failed = false;
    if (!failed) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = st != nullptr && (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone);
        if (!capturing) {
            void* h = nullptr;
            void* d = nullptr;
            if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
                *reinterpret_cast<volatile unsigned int*>(h) = 0u;
                g_lstm_err_host = reinterpret_cast<volatile unsigned int*>(h);
                g_lstm_err_dev = reinterpret_cast<unsigned int*>(d);
                return g_lstm_err_dev;
            }
            failed = true;
            (void)hipGetLastError();
        }
    }
    unsigned int* base = lstm_sync_words();
    g_lstm_fallback_used = true;
    return base ? base + 4 : nullptr;
}
// The word / ticket buffers of the persistent forms (g_lstm_ll, lstm_rs_words, g_lstm_sync) are process-global: two such
// launches must never overlap.  One stream orders its own launches; a launch on ANOTHER stream first waits for an event
// recorded behind the previous persistent launch (LstmTurn below).  Inside a stream capture no event is recorded or waited
// for: the nodes of one graph are ordered by the capture, and graphs that contain these kernels have to be replayed on one
// stream (GraphedStep / SegmentedStep / FlaggedStep replay on the stream they were captured on).
struct LstmTurn {
    hipStream_t st;
    bool eager;
    explicit LstmTurn(hipStream_t s) : st(s), eager(false) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        // (the legacy default stream cannot be captured)
        if (st != nullptr && hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return; }
        eager = cs == hipStreamCaptureStatusNone;
        if (eager && s_event && s_last != st) (void)hipStreamWaitEvent(st, s_event, 0);
    }
    ~LstmTurn() {
        if (!eager) return;
        if (!s_event && hipEventCreateWithFlags(&s_event, hipEventDisableTiming) != hipSuccess) { s_event = nullptr; (void)hipGetLastError(); return; }
        if (hipEventRecord(s_event, st) == hipSuccess) s_last = st; else (void)hipGetLastError();
    }
    static hipEvent_t s_event;
    static hipStream_t s_last;
};
hipEvent_t LstmTurn::s_event = nullptr;
hipStream_t LstmTurn::s_last = nullptr;
int g_lstm_poll_limit = 0;    // icka_lstm_test_hooks: 0 = the per-form defaults below
int g_lstm_test_drop = -1;    // icka_lstm_test_hooks
static int lstm_poll_limit(int dflt) { return g_lstm_poll_limit > 0 ? g_lstm_poll_limit : dflt; }

// W^T for both directions: in [2][R][C] -> out [2][C][R] (bf16), 32x32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R,
                                                        int C) {
    __shared__ bf16_t tile[32][33];
    const int64_t base = (int64_t)blockIdx.z * R * C;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        if (r0 + i < R && c0 + tx < C) tile[i][tx] = in[base + (int64_t)(r0 + i) * C + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) out[base + (int64_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

// -----------------------------------------------------------------------------------------------------------------
// y[M <= 64, N] = act(x . W^T + bias) for a handful of rows (the BERT pooler: tanh(dense(hidden[:, 0])), 32 rows):
// the same operands-from-L2 MFMA scheme as the recurrence -- a block per 16 output columns, the K reduction split over
// its 4 waves, every 16-byte load issued before the MFMAs.  (128x128 GEMM tiles on 32 rows are a latency chain of
// 12 k-tiles on 6 CUs: 35 us.)
struct SmallLinArgs {
    const bf16_t* x; int64_t ldx; const bf16_t* W; const float* bias; bf16_t* y; int64_t ldy;
    int M, N, K, act;   // act: 0 none, 1 tanh
};
template <int NRT>
__global__ __launch_bounds__(256) void linear_small_kernel(const SmallLinArgs a) {
    __shared__ float s_p[4][NRT * 16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, i15 = lane & 15, g4 = lane >> 4;
    const int kq = a.K / 4;   // K % 128 == 0
    f32x4 acc[NRT];
#pragma unroll
    for (int r = 0; r < NRT; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* wrow = a.W + (int64_t)(n0 + i15) * a.K + wave * kq + 8 * g4;
    const bf16_t* xbase = a.x + wave * kq + 8 * g4;
    const bool nok = n0 + i15 < a.N;
    for (int kb = 0; kb < kq; kb += 256) {
        bf16x8 wf[8], xf[NRT][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k0 = kb + 32 * u;
            const bool kok = k0 < kq;
            wf[u] = frag16(wrow + k0, kok && nok);
#pragma unroll
            for (int r = 0; r < NRT; ++r) {
                const int m = 16 * r + i15;
                xf[r][u] = frag16(xbase + (int64_t)m * a.ldx + k0, kok && m < a.M);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < NRT; ++r) acc[r] = mfma16(wf[u], xf[r][u], acc[r]);   // D[i = column 4*g4+q][j = row i15]
    }
#pragma unroll
    for (int r = 0; r < NRT; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) s_p[wave][16 * r + i15][4 * g4 + q] = acc[r][q];
    __syncthreads();
    for (int p = tid; p < NRT * 256; p += 256) {
        const int m = p >> 4, u = p & 15;
        if (m >= a.M || n0 + u >= a.N) continue;
        float v = s_p[0][m][u] + s_p[1][m][u] + s_p[2][m][u] + s_p[3][m][u] + (a.bias ? a.bias[n0 + u] : 0.f);
        if (a.act == 1) v = tanhf(v);
        a.y[(int64_t)m * a.ldy + n0 + u] = f2bf(v);
    }
}

inline int lstm_check(int B, int S, int H) {
    if (B <= 0 || S <= 0 || H <= 0) return ICKA_E_SHAPE;
    if (B > 16 * MAX_RT || H % 32 != 0) return ICKA_E_SHAPE;
    return 0;
}

}  // namespace

extern "C" int icka_lstm_fwd(const float* gates_x, int64_t ldg, const void* w_hh, void* y, float* c_all, void* act,
                             void* hprev, int32_t B, int32_t S, int32_t H, int32_t flags, void* stream) {
    if (!gates_x || !w_hh || !y || !c_all || !act) return ICKA_E_ARG;
    if (flags & ~(ICKA_LSTM_PER_STEP | ICKA_LSTM_TICKETS | ICKA_LSTM_NO_BATCH_SPLIT)) return ICKA_E_ARG;
    const int persistent = !(flags & ICKA_LSTM_PER_STEP), handoff = !(flags & ICKA_LSTM_TICKETS),
              bsplit = !(flags & ICKA_LSTM_NO_BATCH_SPLIT);
    if (int rc = lstm_check(B, S, H)) return rc;
    if (ldg < 8 * (int64_t)H) return ICKA_E_ARG;
    if ((reinterpret_cast<uintptr_t>(w_hh) | reinterpret_cast<uintptr_t>(y)) & 15) return ICKA_E_ALIGN;
    LstmArgs a{};
    a.gx = gates_x; a.ldg = ldg; a.whh = (const bf16_t*)w_hh; a.y = (bf16_t*)y; a.c_all = c_all;
    a.act = (bf16_t*)act; a.hprev = (bf16_t*)hprev; a.B = B; a.S = S; a.H = H;
    a.test_drop = g_lstm_test_drop;
    const int nrt = (B + 15) / 16;
    if (persistent && handoff == 1 && nrt <= 2 && H <= LSTM_LL_MAXH && H % 128 == 0 && (H / 16) * 2 <= lstm_cus()) {
        // flag-in-data hand-off: zero the word buffers (tags of an earlier launch), then one launch for all S steps
        hipStream_t st = (hipStream_t)stream;
        unsigned int* errw = lstm_err_word(st);
        unsigned long long* ll = lstm_ll_words();
        if (!errw || !ll) return ICKA_E_ARG;
        LstmTurn turn(st);
        a.poll_limit = lstm_poll_limit(1 << 20);
        if (hipMemsetAsync(ll, 0, sizeof(unsigned long long) * 2 * 2 * LSTM_LL_ROWS * (LSTM_LL_MAXH / 2), st) != hipSuccess) return ICKA_E_ARG;
        // batch rows are independent recurrences: two tiles of 16 rows run as separate blocks (grid.z) where all of them
        // are co-resident -- half the words to poll and half the MFMAs per block and step
        const bool split = bsplit && nrt == 2 && (H / 16) * 2 * 2 <= lstm_cus();
        const dim3 grid(H / 16, 2, split ? 2 : 1);
        const int nr = split ? 1 : nrt;
#define ICKA_LL_FWD(NRT_, KQ_) hipLaunchKernelGGL((lstm_fwd_ll_kernel<NRT_, KQ_>), grid, dim3(256), 0, st, a, errw)
        switch (H / 128) {
            case 2: if (nr == 1) ICKA_LL_FWD(1, 2); else ICKA_LL_FWD(2, 2); break;
            case 4: if (nr == 1) ICKA_LL_FWD(1, 4); else ICKA_LL_FWD(2, 4); break;
            case 6: if (nr == 1) ICKA_LL_FWD(1, 6); else ICKA_LL_FWD(2, 6); break;
            case 8: if (nr == 1) ICKA_LL_FWD(1, 8); else ICKA_LL_FWD(2, 8); break;
            default:   // H = 128, 384, 640, 896: not instantiated, fall through to the ticket form below
                goto ticket_form;
        }
#undef ICKA_LL_FWD
        ICKA_CHECK_LAUNCH();
        return 0;
    }
ticket_form:
    if (persistent && nrt <= 2 && H <= 1024 && (H / 16) * 2 <= lstm_cus()) {
        LstmPersist ps;
        unsigned int* base = lstm_sync_words();
        unsigned int* errw = lstm_err_word((hipStream_t)stream);
        if (!base || !errw) return ICKA_E_ARG;
        ps.tickets = base; ps.err = errw;
        LstmTurn turn((hipStream_t)stream);
        a.poll_limit = lstm_poll_limit(1 << 22);
        if (hipMemsetAsync(base, 0, 4 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess) return ICKA_E_ARG;
        // KS = resident W_hh fragments per wave (32 hidden units each): 24 up to H = 768, 32 up to H = 1024
        if (H > 768) {
            if (nrt == 1) hipLaunchKernelGGL((lstm_fwd_persistent_kernel<1, 32>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
            else hipLaunchKernelGGL((lstm_fwd_persistent_kernel<2, 32>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
        } else if (nrt == 1) hipLaunchKernelGGL((lstm_fwd_persistent_kernel<1, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
        else hipLaunchKernelGGL((lstm_fwd_persistent_kernel<2, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    for (int s = 0; s < S; ++s) {
        a.step = s;
        if (nrt == 1) hipLaunchKernelGGL((lstm_fwd_step_kernel<1, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a);
        else if (nrt == 2) hipLaunchKernelGGL((lstm_fwd_step_kernel<2, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((lstm_fwd_step_kernel<4, 8>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a);
        ICKA_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int icka_lstm_bwd(const void* dy, const void* w_hh_t, const void* act, const float* c_all, void* dgates,
                             int64_t ldg, float* dc_carry, int32_t B, int32_t S, int32_t H, int32_t flags, void* stream) {
    if (!dy || !w_hh_t || !act || !c_all || !dgates || !dc_carry) return ICKA_E_ARG;
    if (flags & ~(ICKA_LSTM_PER_STEP | ICKA_LSTM_TICKETS | ICKA_LSTM_NO_BATCH_SPLIT)) return ICKA_E_ARG;
    const int persistent = !(flags & ICKA_LSTM_PER_STEP), handoff = !(flags & ICKA_LSTM_TICKETS);
    if (int rc = lstm_check(B, S, H)) return rc;
    if (ldg < 8 * (int64_t)H || ldg % 8) return ICKA_E_ARG;
    if ((reinterpret_cast<uintptr_t>(w_hh_t) | reinterpret_cast<uintptr_t>(dgates)) & 15) return ICKA_E_ALIGN;
    LstmArgs a{};
    a.dy = (const bf16_t*)dy; a.whh = (const bf16_t*)w_hh_t; a.act = (bf16_t*)const_cast<void*>(act);
    a.c_all = const_cast<float*>(c_all); a.dgates = (bf16_t*)dgates; a.ldg = ldg; a.dc_carry = dc_carry;
    a.B = B; a.S = S; a.H = H;
    a.test_drop = g_lstm_test_drop;
    const int nrt = (B + 15) / 16;
    if (persistent && handoff == 1 && nrt <= 2 && H <= LSTM_LL_MAXH && H % 256 == 0 && (H / 16) * 2 * nrt <= lstm_cus()) {
        // reduce-scatter form with tagged words (lstm_bwd_rs_kernel); the word buffer is allocated once (largest shape)
        unsigned int* errw = lstm_err_word((hipStream_t)stream);
        unsigned long long* llr = lstm_rs_words((hipStream_t)stream);
        if (errw && llr) {
            hipStream_t st = (hipStream_t)stream;
            LstmTurn turn(st);
            a.poll_limit = lstm_poll_limit(1 << 17);
            const int nblk = H / 16;
            if (hipMemsetAsync(llr, 0, sizeof(unsigned long long) * 4 * nblk * nblk * 256, st) != hipSuccess) return ICKA_E_ARG;
            const dim3 grid(nblk, 2, nrt);
            switch (H / 256) {
                case 1: hipLaunchKernelGGL((lstm_bwd_rs_kernel<4>), grid, dim3(256), 0, st, a, llr, errw); break;
                case 2: hipLaunchKernelGGL((lstm_bwd_rs_kernel<8>), grid, dim3(256), 0, st, a, llr, errw); break;
                case 3: hipLaunchKernelGGL((lstm_bwd_rs_kernel<12>), grid, dim3(256), 0, st, a, llr, errw); break;
                default: hipLaunchKernelGGL((lstm_bwd_rs_kernel<16>), grid, dim3(256), 0, st, a, llr, errw); break;
            }
            ICKA_CHECK_LAUNCH();
            return 0;
        }
    }
    if (persistent && nrt <= 2 && H <= 1024 && (H / 16) * 2 <= lstm_cus()) {
        LstmPersist ps;
        unsigned int* base = lstm_sync_words();
        unsigned int* errw = lstm_err_word((hipStream_t)stream);
        if (!base || !errw) return ICKA_E_ARG;
        ps.tickets = base; ps.err = errw;
        LstmTurn turn((hipStream_t)stream);
        a.poll_limit = lstm_poll_limit(1 << 22);
        if (hipMemsetAsync(base, 0, 4 * sizeof(unsigned int), (hipStream_t)stream) != hipSuccess) return ICKA_E_ARG;
        // (batch tiles as separate blocks, as in the forward launch, are slower here: 9.3 vs 8.9 us per step at H = 768 --
        //  every block reads all of dgates_t through L2 either way, twice the blocks only add contention)
        if (H > 768) {
            if (nrt == 1) hipLaunchKernelGGL((lstm_bwd_persistent_kernel<1, 32>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
            else hipLaunchKernelGGL((lstm_bwd_persistent_kernel<2, 32>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
        } else if (nrt == 1) hipLaunchKernelGGL((lstm_bwd_persistent_kernel<1, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
        else hipLaunchKernelGGL((lstm_bwd_persistent_kernel<2, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a, ps);
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    for (int s = S - 1; s >= 0; --s) {
        a.step = s;
        if (nrt == 1) hipLaunchKernelGGL((lstm_bwd_step_kernel<1, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a);
        else if (nrt == 2) hipLaunchKernelGGL((lstm_bwd_step_kernel<2, 24>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((lstm_bwd_step_kernel<4, 8>), dim3(H / 16, 2), dim3(256), 0, (hipStream_t)stream, a);
        ICKA_CHECK_LAUNCH();
    }
    return 0;
}

extern "C" int icka_transpose_bf16(const void* in, void* out, int32_t batch, int32_t R, int32_t C, void* stream) {
    if (!in || !out) return ICKA_E_ARG;
    if (batch <= 0 || R <= 0 || C <= 0) return ICKA_E_SHAPE;
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)in, (bf16_t*)out, R, C);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_linear_small_m(const void* x, int64_t ldx, const void* W, const float* bias, void* y, int64_t ldy,
                                   int32_t M, int32_t N, int32_t K, int32_t act, void* stream) {
    if (!x || !W || !y) return ICKA_E_ARG;
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || K % 128 != 0 || act < 0 || act > 1) return ICKA_E_SHAPE;
    if (ldx % 8 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15)) return ICKA_E_ALIGN;
    const SmallLinArgs a{(const bf16_t*)x, ldx, (const bf16_t*)W, bias, (bf16_t*)y, ldy, M, N, K, act};
    const int grid = (N + 15) / 16, nrt = (M + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    if (nrt == 1) hipLaunchKernelGGL((linear_small_kernel<1>), dim3(grid), dim3(256), 0, st, a);
    else if (nrt == 2) hipLaunchKernelGGL((linear_small_kernel<2>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((linear_small_kernel<4>), dim3(grid), dim3(256), 0, st, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

/* 1 if a hand-off wait of a persistent recurrence ever gave up (the outputs of that call are NaN-poisoned).  Reads the
   host-mapped error word when there is one (mapped by the first persistent launch issued outside a stream capture) -- a plain
   host read, no runtime call; only launches that had to take the fallback device word cost a blocking copy here. */
extern "C" int icka_lstm_barrier_error(void) {
    if (g_lstm_err_host) return (int)*g_lstm_err_host;
    if (!g_lstm_fallback_used) return 0;   // no persistent launch has raised into the device word: nothing to read (and no
                                           // runtime call from here, so the query is legal inside a stream capture)
    unsigned int w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(w, HIP_SYMBOL(g_lstm_sync), sizeof(w)) != hipSuccess) return -1;
    return (int)w[4];
}
/* Reset the error word(s) after the caller has handled (raised) the error. */
extern "C" int icka_lstm_clear_error(void) {
    if (g_lstm_err_host) *g_lstm_err_host = 0u;
    unsigned int* base = lstm_sync_words();
    if (base && hipMemset(base + 4, 0, sizeof(unsigned int)) != hipSuccess) return ICKA_E_ARG;
    return 0;
}
extern "C" int icka_lstm_set_reserved_cus(int32_t n) {
    if (n < 0) return ICKA_E_ARG;
    g_icka_reserved_cus = n;
    return 0;
}
/* Test hooks of the give-up path: poll_limit > 0 replaces the per-form poll budgets (0 restores them); drop_step >= 0 makes
   block (0, direction 0, batch tile 0) of the persistent launches skip publishing that step, so its consumers time out. */
extern "C" int icka_lstm_test_hooks(int32_t poll_limit, int32_t drop_step) {
    g_lstm_poll_limit = poll_limit > 0 ? poll_limit : 0;
    g_lstm_test_drop = drop_step >= 0 ? drop_step : -1;
    return 0;
}
