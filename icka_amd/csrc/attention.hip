// Fused multi-head attention for the ICKA hot path (gfx950), head size 64, self- and cross-attention
// (BertSelfAttention.forward Cross_Modal_Interaction_Module.py:478-506, BertCoAttention.forward :590-624):
//     S = Q.K^T ; S *= 1/sqrt(d) ; S += additive mask ; P = softmax(S) ; P = dropout(P) ; O = P.V
// Forward: one block per (batch, head, 64-query tile), 4 waves x 16 queries; K/V tiles of 64 keys staged through
// LDS; QK^T and PV on v_mfma_f32_16x16x32_bf16.  Scores are computed TRANSPOSED (S^T = K.Q^T): the query sits on
// the lane (l&15) and 4 consecutive keys in the accumulator registers, so (a) the softmax row reduction is in-lane
// plus two cross-lane shuffles, and (b) the bf16-packed probabilities ARE the B operand of O^T = V^T.P^T with no LDS
// round trip; V^T fragments come from the row-major V tile through ds_read_b64_tr_b16.  Online softmax over key
// tiles; only the per-row log-sum-exp is saved.
// Backward recomputes P from (Q, K, lse) in two kernels with the same structure and no cross-block reduction:
//   dq kernel  (block owns 64 queries, sweeps keys):   dS^T -> dQ^T = K^T.dS^T
//   dkv kernel (block owns 64 keys, sweeps queries):   dV^T = dO^T.Pd ,  dK^T = Q^T.dS
// The dropout mask is re-generated from the counter hash of the element index (common.h), identical in all three.
#include <math.h>

#include "common.h"

namespace {

constexpr int HD = 64;           // head size
constexpr int TILE = 64;         // rows per LDS tile
constexpr int TILE_B = TILE * HD * 2;

// [64 rows][64 d] bf16 tile image serving both row reads (ds_read_b128) and transposed reads (ds_read_b64_tr_b16):
// 8-row x 32-col sub-tiles of 512 B, 16-B chunk XORed with (row>>2)&3 (cdna guide T10, image (a)).
__device__ __forceinline__ uint32_t off_t(int row, int ch) {
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}

// Stage rows [row0, row0+64) of a token-major matrix (this head's 64 columns) into an LDS tile; rows >= nrows -> 0.
__device__ __forceinline__ void stage_tile(char* lds, const bf16_t* base, int64_t ld, int row0, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = tid + 256 * i, r = q >> 3, c = q & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + r < nrows) v = *reinterpret_cast<const u32x4*>(base + (int64_t)(row0 + r) * ld + c * 8);
        *reinterpret_cast<u32x4*>(lds + off_t(r, c)) = v;
    }
}

// row-read fragment: element e = tile[rbase + (l&15)][32*ks + 8*(l>>4) + e]
__device__ __forceinline__ bf16x8 frag_row(const char* tile, int rbase, int ks, int lane) {
    return lds_read_b128(tile, off_t(rbase + (lane & 15), 4 * ks + (lane >> 4)));
}
// transposed fragment for the "accumulator as B operand" k-slot order: lane l, element e (0..7) =
//   tile[row = 32*ks + 16*(e>>2) + 4*(l>>4) + (e&3)][col = 16*dt + (l&15)]
__device__ __forceinline__ bf16x8 frag_tr(const char* tile, int dt, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row = 32 * ks + 4 * g + q, ch = 2 * dt + (p >> 1);
    const bf16x4 lo = lds_read_tr(tile, off_t(row, ch) + 8 * (p & 1));
    const bf16x4 hi = lds_read_tr(tile, off_t(row + 16, ch) + 8 * (p & 1));
    return join8(lo, hi);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
    bf16x8 o = {f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
    return o;
}

struct AttnArgs {
    const bf16_t* Q; int64_t ldq; const bf16_t* K; int64_t ldk; const bf16_t* V; int64_t ldv;
    const float* mask; const bf16_t* O; int64_t ldo; const bf16_t* dO; int64_t lddo;
    bf16_t* Ow; float* lse; float* delta;
    bf16_t* dQ; int64_t lddq; bf16_t* dK; int64_t lddk; bf16_t* dV; int64_t lddv;
    int B, h, Sq, Skv; float scale; DropCfg drop;
};

// ------------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a_) {
    AttnArgs a = a_;
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char smem[3 * TILE_B];
    char* sQ = smem; char* sK = smem + TILE_B; char* sV = smem + 2 * TILE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i15 = lane & 15;
    const int nqt = (a.Sq + TILE - 1) / TILE;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, head = bh % a.h, b = bh / a.h;
    const bf16_t* Qb = a.Q + (int64_t)b * a.Sq * a.ldq + head * HD;
    const bf16_t* Kb = a.K + (int64_t)b * a.Skv * a.ldk + head * HD;
    const bf16_t* Vb = a.V + (int64_t)b * a.Skv * a.ldv + head * HD;
    const float* mb = a.mask + (int64_t)b * a.Skv;

    stage_tile(sQ, Qb, a.ldq, qt * TILE, a.Sq, tid);
    __syncthreads();
    bf16x8 qf[2];
    qf[0] = frag_row(sQ, 16 * wave, 0, lane);
    qf[1] = frag_row(sQ, 16 * wave, 1, lane);
    const int q = qt * TILE + 16 * wave + i15;
    const uint32_t idx_row = ((uint32_t)(b * a.h + head) * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc_o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc_o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kv0 = 0; kv0 < a.Skv; kv0 += TILE) {
        if (kv0) __syncthreads();
        stage_tile(sK, Kb, a.ldk, kv0, a.Skv, tid);
        stage_tile(sV, Vb, a.ldv, kv0, a.Skv, tid);
        __syncthreads();

        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) s[kt] = mfma16(frag_row(sK, 16 * kt, ks, lane), qf[ks], s[kt]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 16 * kt + 4 * g + r;
                const float v = key < a.Skv ? s[kt][r] * a.scale + mb[key] : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m_new);
                psum += p;
                const int key = kv0 + 16 * kt + 4 * g + r;
                s[kt][r] = p * drop_mul(a.drop, idx_row + (uint32_t)key);
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        bf16x8 pf[2];
        pf[0] = pack8(s[0], s[1]);
        pf[1] = pack8(s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            acc_o[dt] *= alpha;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc_o[dt] = mfma16(frag_tr(sV, dt, ks, lane), pf[ks], acc_o[dt]);
        }
    }
    if (q < a.Sq) {
        const float inv = 1.f / l_run;
        bf16_t* orow = a.Ow + ((int64_t)b * a.Sq + q) * a.ldo + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<u32x2*>(orow + 16 * dt + 4 * g) =
                pack4(acc_o[dt][0] * inv, acc_o[dt][1] * inv, acc_o[dt][2] * inv, acc_o[dt][3] * inv);
        if (g == 0 && a.lse) a.lse[(int64_t)(b * a.h + head) * a.Sq + q] = m_run + logf(l_run);
    }
}

// ------------------------------------------------------------------------------------------- delta = rowsum(dO*O)
__global__ void attn_delta_kernel(const bf16_t* __restrict__ O, int64_t ldo, const bf16_t* __restrict__ dO,
                                  int64_t lddo, float* __restrict__ delta, int B, int h, int Sq) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per 8-element chunk
    const int cpr = h * 8;                                               // chunks per token row
    const int64_t total = (int64_t)B * Sq * cpr;
    float s = 0.f;
    int64_t row = 0; int c = 0;
    if (idx < total) {
        row = idx / cpr; c = (int)(idx - row * cpr);
        const bf16x8 o = as_bf16x8(*reinterpret_cast<const u32x4*>(O + row * ldo + c * 8));
        const bf16x8 d = as_bf16x8(*reinterpret_cast<const u32x4*>(dO + row * lddo + c * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) s += bf2f(o[e]) * bf2f(d[e]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (idx < total && (c & 7) == 0) {
        const int head = c >> 3;
        const int64_t b = row / Sq, sq = row - b * Sq;
        delta[(b * h + head) * Sq + sq] = s;
    }
}

// ---------------------------------------------------------------------------------------------------- dQ kernel
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const AttnArgs a_) {
    AttnArgs a = a_;
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];
    char* sQ = smem; char* sDO = smem + TILE_B; char* sK = smem + 2 * TILE_B; char* sV = smem + 3 * TILE_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i15 = lane & 15;
    const int nqt = (a.Sq + TILE - 1) / TILE;
    const int qt = blockIdx.x % nqt, bh = blockIdx.x / nqt, head = bh % a.h, b = bh / a.h;
    const bf16_t* Qb = a.Q + (int64_t)b * a.Sq * a.ldq + head * HD;
    const bf16_t* dOb = a.dO + (int64_t)b * a.Sq * a.lddo + head * HD;
    const bf16_t* Kb = a.K + (int64_t)b * a.Skv * a.ldk + head * HD;
    const bf16_t* Vb = a.V + (int64_t)b * a.Skv * a.ldv + head * HD;
    const float* mb = a.mask + (int64_t)b * a.Skv;

    stage_tile(sQ, Qb, a.ldq, qt * TILE, a.Sq, tid);
    stage_tile(sDO, dOb, a.lddo, qt * TILE, a.Sq, tid);
    __syncthreads();
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        qf[ks] = frag_row(sQ, 16 * wave, ks, lane);
        dof[ks] = frag_row(sDO, 16 * wave, ks, lane);
    }
    const int q = qt * TILE + 16 * wave + i15;
    const bool qok = q < a.Sq;
    const int64_t stat = (int64_t)(b * a.h + head) * a.Sq + q;
    const float lse_q = qok ? a.lse[stat] : INFINITY;
    const float dl_q = qok ? a.delta[stat] : 0.f;
    const uint32_t idx_row = ((uint32_t)(b * a.h + head) * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;

    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kv0 = 0; kv0 < a.Skv; kv0 += TILE) {
        if (kv0) __syncthreads();
        stage_tile(sK, Kb, a.ldk, kv0, a.Skv, tid);
        stage_tile(sV, Vb, a.ldv, kv0, a.Skv, tid);
        __syncthreads();
        f32x4 ds[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16(frag_row(sK, 16 * kt, ks, lane), qf[ks], s);
                dp = mfma16(frag_row(sV, 16 * kt, ks, lane), dof[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 16 * kt + 4 * g + r;
                const float p = key < a.Skv ? __expf(s[r] * a.scale + mb[key] - lse_q) : 0.f;
                const float dm = drop_mul(a.drop, idx_row + (uint32_t)key);
                ds[kt][r] = p * (dp[r] * dm - dl_q) * a.scale;
            }
        }
        bf16x8 dsf[2];
        dsf[0] = pack8(ds[0], ds[1]);
        dsf[1] = pack8(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc[dt] = mfma16(frag_tr(sK, dt, ks, lane), dsf[ks], acc[dt]);
    }
    if (qok) {
        bf16_t* row = a.dQ + ((int64_t)b * a.Sq + q) * a.lddq + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<u32x2*>(row + 16 * dt + 4 * g) = pack4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
    }
}

// -------------------------------------------------------------------------------------------------- dK/dV kernel
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const AttnArgs a_) {
    AttnArgs a = a_;
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B + 2 * TILE * 4];
    char* sK = smem; char* sV = smem + TILE_B; char* sQ = smem + 2 * TILE_B; char* sDO = smem + 3 * TILE_B;
    float* s_lse = reinterpret_cast<float*>(smem + 4 * TILE_B);
    float* s_dl = s_lse + TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i15 = lane & 15;
    const int nkt = (a.Skv + TILE - 1) / TILE;
    const int kt_blk = blockIdx.x % nkt, bh = blockIdx.x / nkt, head = bh % a.h, b = bh / a.h;
    const bf16_t* Qb = a.Q + (int64_t)b * a.Sq * a.ldq + head * HD;
    const bf16_t* dOb = a.dO + (int64_t)b * a.Sq * a.lddo + head * HD;
    const bf16_t* Kb = a.K + (int64_t)b * a.Skv * a.ldk + head * HD;
    const bf16_t* Vb = a.V + (int64_t)b * a.Skv * a.ldv + head * HD;

    stage_tile(sK, Kb, a.ldk, kt_blk * TILE, a.Skv, tid);
    stage_tile(sV, Vb, a.ldv, kt_blk * TILE, a.Skv, tid);
    __syncthreads();
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = frag_row(sK, 16 * wave, ks, lane);
        vf[ks] = frag_row(sV, 16 * wave, ks, lane);
    }
    const int key = kt_blk * TILE + 16 * wave + i15;
    const bool kok = key < a.Skv;
    const float mk = kok ? a.mask[(int64_t)b * a.Skv + key] : 0.f;
    const uint32_t idx_bh = (uint32_t)(b * a.h + head) * (uint32_t)a.Sq;

    f32x4 acc_k[4], acc_v[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { acc_k[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; acc_v[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    for (int q0 = 0; q0 < a.Sq; q0 += TILE) {
        __syncthreads();
        stage_tile(sQ, Qb, a.ldq, q0, a.Sq, tid);
        stage_tile(sDO, dOb, a.lddo, q0, a.Sq, tid);
        if (tid < TILE) {
            const int qq = q0 + tid;
            const int64_t stat = (int64_t)(b * a.h + head) * a.Sq + qq;
            s_lse[tid] = qq < a.Sq ? a.lse[stat] : INFINITY;
            s_dl[tid] = qq < a.Sq ? a.delta[stat] : 0.f;
        }
        __syncthreads();
        f32x4 pd[4], ds[4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16(frag_row(sQ, 16 * qt, ks, lane), kf[ks], s);
                dp = mfma16(frag_row(sDO, 16 * qt, ks, lane), vf[ks], dp);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * qt + 4 * g + r;
                const float p = kok ? __expf(s[r] * a.scale + mk - s_lse[ql]) : 0.f;
                const uint32_t idx = (idx_bh + (uint32_t)(q0 + ql)) * (uint32_t)a.Skv + (uint32_t)key;
                const float dm = drop_mul(a.drop, idx);
                pd[qt][r] = p * dm;
                ds[qt][r] = p * (dp[r] * dm - s_dl[ql]) * a.scale;
            }
        }
        bf16x8 pdf[2], dsf[2];
        pdf[0] = pack8(pd[0], pd[1]); pdf[1] = pack8(pd[2], pd[3]);
        dsf[0] = pack8(ds[0], ds[1]); dsf[1] = pack8(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc_v[dt] = mfma16(frag_tr(sDO, dt, ks, lane), pdf[ks], acc_v[dt]);
                acc_k[dt] = mfma16(frag_tr(sQ, dt, ks, lane), dsf[ks], acc_k[dt]);
            }
    }
    if (kok) {
        bf16_t* krow = a.dK + ((int64_t)b * a.Skv + key) * a.lddk + head * HD;
        bf16_t* vrow = a.dV + ((int64_t)b * a.Skv + key) * a.lddv + head * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *reinterpret_cast<u32x2*>(krow + 16 * dt + 4 * g) =
                pack4(acc_k[dt][0], acc_k[dt][1], acc_k[dt][2], acc_k[dt][3]);
            *reinterpret_cast<u32x2*>(vrow + 16 * dt + 4 * g) =
                pack4(acc_v[dt][0], acc_v[dt][1], acc_v[dt][2], acc_v[dt][3]);
        }
    }
}

inline bool ok16(const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; }

}  // namespace

extern "C" int icka_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                             const float* add_mask, void* O, int64_t ldo, float* lse, int32_t B, int32_t heads,
                             int32_t Sq, int32_t Skv, float scale, float p_drop, uint64_t seed, void* stream) {
    if (!Q || !K || !V || !add_mask || !O) return ICKA_E_ARG;
    if (B <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0) return ICKA_E_SHAPE;
    if ((int64_t)B * heads * Sq * Skv >= (1ll << 32)) return ICKA_E_SHAPE;  // 32-bit dropout counter
    if (!ok16(Q, ldq) || !ok16(K, ldk) || !ok16(V, ldv) || !ok16(O, ldo)) return ICKA_E_ALIGN;
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
    a.mask = add_mask; a.Ow = (bf16_t*)O; a.ldo = ldo; a.lse = lse;
    a.B = B; a.h = heads; a.Sq = Sq; a.Skv = Skv; a.scale = scale; a.drop = make_drop(p_drop, seed);
    const int grid = B * heads * ((Sq + TILE - 1) / TILE);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                             const float* add_mask, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                             const float* lse, float* delta, void* dQ, int64_t lddq, void* dK, int64_t lddk,
                             void* dV, int64_t lddv, int32_t B, int32_t heads, int32_t Sq, int32_t Skv, float scale,
                             float p_drop, uint64_t seed, void* stream) {
    if (!Q || !K || !V || !add_mask || !O || !dO || !lse || !delta || !dQ || !dK || !dV) return ICKA_E_ARG;
    if (B <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0) return ICKA_E_SHAPE;
    if ((int64_t)B * heads * Sq * Skv >= (1ll << 32)) return ICKA_E_SHAPE;
    if (!ok16(Q, ldq) || !ok16(K, ldk) || !ok16(V, ldv) || !ok16(O, ldo) || !ok16(dO, lddo) || !ok16(dQ, lddq) ||
        !ok16(dK, lddk) || !ok16(dV, lddv))
        return ICKA_E_ALIGN;
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
    a.mask = add_mask; a.O = (const bf16_t*)O; a.ldo = ldo; a.dO = (const bf16_t*)dO; a.lddo = lddo;
    a.lse = const_cast<float*>(lse); a.delta = delta;
    a.dQ = (bf16_t*)dQ; a.lddq = lddq; a.dK = (bf16_t*)dK; a.lddk = lddk; a.dV = (bf16_t*)dV; a.lddv = lddv;
    a.B = B; a.h = heads; a.Sq = Sq; a.Skv = Skv; a.scale = scale; a.drop = make_drop(p_drop, seed);
    hipStream_t st = (hipStream_t)stream;
    const int64_t chunks = (int64_t)B * Sq * heads * 8;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, a.O, ldo, a.dO,
                       lddo, delta, B, heads, Sq);
    ICKA_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * heads * ((Sq + TILE - 1) / TILE)), dim3(256), 0, st, a);
    ICKA_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * heads * ((Skv + TILE - 1) / TILE)), dim3(256), 0, st, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}
