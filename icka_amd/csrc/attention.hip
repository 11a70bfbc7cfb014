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
#include "attn_core.h"

namespace {

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
        for (int kt = 0; kt < 4; ++kt) {
            float dm4[4];
            drop_mul_key4(a.drop, idx_row, (uint32_t)(kv0 + 16 * kt + 4 * g), dm4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __expf(s[kt][r] - m_new);
                psum += p;
                s[kt][r] = p * dm4[r];
            }
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
        if (a.Ow16) {
            _Float16* hrow = a.Ow16 + ((int64_t)b * a.Sq + q) * a.ldo + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<f16x4*>(hrow + 16 * dt + 4 * g) =
                    f16x4{(_Float16)(acc_o[dt][0] * inv), (_Float16)(acc_o[dt][1] * inv), (_Float16)(acc_o[dt][2] * inv),
                          (_Float16)(acc_o[dt][3] * inv)};
        }
        if (g == 0 && a.lse) a.lse[(int64_t)(b * a.h + head) * a.Sq + q] = m_run + logf(l_run);
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
    const uint32_t idx_row = ((uint32_t)(b * a.h + head) * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;

    // Pass 1: delta_q = sum_j P_qj * dP_qj in f32 from the recomputed P and dP.  The flash-attention shortcut
    // delta = rowsum(dO * O) reads the bf16-ROUNDED output: its 2^-9 relative error is harmless per se, but dS =
    // P * (dP - delta) cancels almost completely once the value rows of a head resemble each other (deep layers), and
    // the query / key weight gradients of the last bert-large layers then came out 10 % off (tests/test_fullsize_gpu.py).
    float dl_q = 0.f;
    for (int kv0 = 0; kv0 < a.Skv; kv0 += TILE) {
        if (kv0) __syncthreads();
        stage_tile(sK, Kb, a.ldk, kv0, a.Skv, tid);
        stage_tile(sV, Vb, a.ldv, kv0, a.Skv, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16(frag_row(sK, 16 * kt, ks, lane), qf[ks], s);
                dp = mfma16(frag_row(sV, 16 * kt, ks, lane), dof[ks], dp);
            }
            float dm4[4];
            drop_mul_key4(a.drop, idx_row, (uint32_t)(kv0 + 16 * kt + 4 * g), dm4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 16 * kt + 4 * g + r;
                const float p = key < a.Skv ? __expf(s[r] * a.scale + mb[key] - lse_q) : 0.f;
                dl_q += p * dp[r] * dm4[r];
            }
        }
    }
    dl_q += __shfl_xor(dl_q, 16, 64);
    dl_q += __shfl_xor(dl_q, 32, 64);
    if (g == 0 && qok) a.delta[stat] = dl_q;    // read by attn_bwd_dkv_kernel (launched after this kernel)

    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kv0 = 0; kv0 < a.Skv; kv0 += TILE) {
        __syncthreads();
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
            float dm4[4];
            drop_mul_key4(a.drop, idx_row, (uint32_t)(kv0 + 16 * kt + 4 * g), dm4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 16 * kt + 4 * g + r;
                const float p = key < a.Skv ? __expf(s[r] * a.scale + mb[key] - lse_q) : 0.f;
                ds[kt][r] = p * (dp[r] * dm4[r] - dl_q) * a.scale;
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

// dQ kernel for Skv <= 4 * 64 keys (bert-large at seq 256, config c4): the first pass keeps P and dP of ALL keys of this
// wave's 16 queries in registers (2 x NKT x 16 floats per lane) and the K tiles resident in LDS, so the second pass only
// forms dS = P * (dP - delta) and multiplies it into dQ -- no second QK^T / dO.V^T, no re-staging (the two-pass kernel
// above recomputes both: 158 us -> per launch at B32 x 16 heads x 256 x 256, 12.5 % of the c4 step).
template <int NKT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_keep_kernel(const AttnArgs a_) {
    AttnArgs a = a_;
    a.drop = drop_resolve(a.drop);
    __shared__ __attribute__((aligned(16))) char smem[(3 + NKT) * TILE_B];
    char* sQ = smem; char* sDO = smem + TILE_B; char* sV = smem + 2 * TILE_B; char* sK = smem + 3 * TILE_B;
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
    const uint32_t idx_row = ((uint32_t)(b * a.h + head) * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;

    // P is kept bf16-packed (dS is rounded to bf16 for the MFMA anyway; dP - delta, where the cancellation is, stays f32):
    // 2 x NKT x 16 floats pushed the 4-tile instance over 256 VGPRs = one wave per SIMD
    u32x2 pr[NKT][4];
    f32x4 dpr[NKT][4];
    float dl_q = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        const int kv0 = t * TILE;
        if (t) __syncthreads();                      // every wave is done with the previous V tile
        stage_tile(sK + t * TILE_B, Kb, a.ldk, kv0, a.Skv, tid);
        stage_tile(sV, Vb, a.ldv, kv0, a.Skv, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                s = mfma16(frag_row(sK + t * TILE_B, 16 * kt, ks, lane), qf[ks], s);
                dp = mfma16(frag_row(sV, 16 * kt, ks, lane), dof[ks], dp);
            }
            float pv[4], dm4[4];
            drop_mul_key4(a.drop, idx_row, (uint32_t)(kv0 + 16 * kt + 4 * g), dm4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + 16 * kt + 4 * g + r;
                const float p = key < a.Skv ? __expf(s[r] * a.scale + mb[key] - lse_q) : 0.f;
                const float d = dp[r] * dm4[r];
                pv[r] = p;
                dpr[t][kt][r] = d;
                dl_q += p * d;
            }
            pr[t][kt] = pack4(pv[0], pv[1], pv[2], pv[3]);
        }
    }
    dl_q += __shfl_xor(dl_q, 16, 64);
    dl_q += __shfl_xor(dl_q, 32, 64);
    if (g == 0 && qok) a.delta[stat] = dl_q;    // read by attn_bwd_dkv_kernel (launched after this kernel)

    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        f32x4 ds[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const bf16x4 pb = as_bf16x4(pr[t][kt]);
#pragma unroll
            for (int r = 0; r < 4; ++r) ds[kt][r] = bf2f(pb[r]) * (dpr[t][kt][r] - dl_q) * a.scale;
        }
        bf16x8 dsf[2];
        dsf[0] = pack8(ds[0], ds[1]);
        dsf[1] = pack8(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc[dt] = mfma16(frag_tr(sK + t * TILE_B, dt, ks, lane), dsf[ks], acc[dt]);
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
                const float dm = a.drop.thr ? drop_mul_key(a.drop, (idx_bh + (uint32_t)(q0 + ql)) * (uint32_t)a.Skv, (uint32_t)key) : a.drop.scale;
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

// ============================================================================================================
// Whole-head kernels for short sequences (Sq <= 128 and Skv <= 128: the reference's max_seq_length 128 text and its
// 36/49 image regions).  ONE block per (batch, head) stages Q, K, V (and dO) once; every score of the head lives in
// registers, so the forward needs no online-softmax rescaling and the backward computes dQ (waves own queries) and
// dK/dV (waves own keys) from the same LDS tiles in one launch, with delta = rowsum(P.dP) taken from registers
// instead of a separate pass over O.  QT = 16-query sub-tiles per wave (block covers 64*QT queries), KT = 16-key
// sub-tiles in total.
#ifndef ICKA_ATTN_ABLATE
#define ICKA_ATTN_ABLATE 0   // diagnostic builds (tools/attn_bench.py): 1 = no global loads, 2 = no compute
#endif
template <int ROWS>
__device__ __forceinline__ void stage_rows(char* lds, const bf16_t* base, int64_t ld, int nrows, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / 256; ++i) {
        const int q = tid + 256 * i, r = q >> 3, c = q & 7;
        // branch-free: rows past the end re-read the last row and are zeroed by a select (nrows >= 1)
        const int rr = r < nrows ? r : nrows - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ICKA_ATTN_ABLATE != 1) v = *reinterpret_cast<const u32x4*>(base + (int64_t)rr * ld + c * 8);
        const uint32_t keep = r < nrows ? 0xffffffffu : 0u;
        *reinterpret_cast<u32x4*>(lds + off_t(r, c)) = v & keep;
    }
}

// keep bits for a forward that did not take the whole-head kernel (tiled path): one thread per word
__global__ void attn_keepbits_kernel(uint32_t* __restrict__ out, int64_t rows, int Skv, DropCfg d_) {
    const DropCfg d = drop_resolve(d_);
    const int wpl = keep_wpl(Skv);
    const int64_t n = rows * 4 * wpl;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % wpl), g = (int)((i / wpl) & 3);
        const int64_t row = i / (4 * wpl);
        const uint32_t hx = ((uint32_t)row * (uint32_t)Skv + 2u * (uint32_t)g) * ICKA_HASH_C0 + d.s0;
        uint32_t word = 0u;
#pragma unroll
        for (int k8 = 0; k8 < 8; ++k8) {
            const int kt = 8 * w + k8;
            word |= (d.thr ? drop_nibble_x<true>(d, hx + (uint32_t)(8 * kt) * ICKA_HASH_C0) : 0xFu) << (4 * k8);
        }
        out[i] = word;
    }
}

template <int QT, int KT, bool DROP, bool FP8 = false, bool KB = false>
__global__ __launch_bounds__(256) void attn_fwd_small_kernel(const AttnArgs a_) {
    static_assert(DROP || !KB, "keep bits exist with dropout only");
    AttnArgs a = a_;
    a.drop = drop_resolve(a.drop);
    constexpr int QR = 64 * QT, KR = 16 * KT;
    __shared__ __attribute__((aligned(16))) char smem[(QR + 2 * KR) * 128];
    char* sQ = smem; char* sK = smem + QR * 128; char* sV = sK + KR * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, head = bh % a.h, b = bh / a.h;
    const int q0 = blockIdx.y * QR;   // query block of this head (grid.y > 1 splits a head for load balance)
    stage_rows<QR>(sQ, a.Q + ((int64_t)b * a.Sq + q0) * a.ldq + head * HD, a.ldq, a.Sq - q0, tid);
    stage_rows<KR>(sK, a.K + (int64_t)b * a.Skv * a.ldk + head * HD, a.ldk, a.Skv, tid);
    stage_rows<KR>(sV, a.V + (int64_t)b * a.Skv * a.ldv + head * HD, a.ldv, a.Skv, tid);
    __syncthreads();
    if (ICKA_ATTN_ABLATE == 2) {   // staging + stores only
        for (int i = 0; i < QR * 8 / 256; ++i) {
            const int q = tid + 256 * i, r = q >> 3, c = q & 7;
            if (q0 + r < a.Sq) *reinterpret_cast<u32x4*>(a.Ow + ((int64_t)b * a.Sq + q0 + r) * a.ldo + head * HD + c * 8) =
                *reinterpret_cast<const u32x4*>(sQ + off_t(r, c)) ^ *reinterpret_cast<const u32x4*>(sK + off_t(r % KR, c)) ^
                *reinterpret_cast<const u32x4*>(sV + off_t(r % KR, c));
        }
        return;
    }

    attn_fwd_whole_head<QT, KT, DROP, FP8, KB>(a, sQ, sK, sV, QT * wave, q0, bh, b, head, lane);
}

// Heads of up to 256 x 256 (bert-large at seq 256: BASELINE config c4) run as <QT = 4, KT = 16>: one block per CU with
// the whole 512-register file per wave (Pd and dS of a wave's 64 queries x 256 keys stay packed in 256 registers), the
// mask read from LDS instead of registers, and the phase-B exchange done in groups of NCH 64-key chunks that fit the
// dead K/V region (the [QR x KR] matrix no longer does).
template <int QT, int KT, bool DROP, bool KB = false>
__global__ __launch_bounds__(256, (QT * KT >= 36 || QT >= 4 ? 1 : 2)) void attn_bwd_small_kernel(const AttnArgs a_) {
    static_assert(DROP || !KB, "keep bits exist with dropout only");
    AttnArgs a = a_;
    a.drop = drop_resolve(a.drop);
    constexpr int QR = 64 * QT, KR = 16 * KT, KW = KT / 4;
    // exchange region X (= the K/V tiles, dead after phase A, grown to one chunk when they are smaller): NCH chunks of
    // [QR queries x 64 keys] bf16 at a time, NG groups per pass
    constexpr int KV_BYTES = 2 * KR * 128, CH_BYTES = QR * 128;
    constexpr int X_BYTES = KV_BYTES > CH_BYTES ? KV_BYTES : CH_BYTES;
    constexpr int NCAP = (X_BYTES / CH_BYTES) < KW ? (X_BYTES / CH_BYTES) : KW;
    constexpr int NCH = KW % NCAP == 0 ? NCAP : 1;   // chunks per group: the largest that fits and divides KW (else one)
    constexpr int NG = KW / NCH;
    static_assert(KW % NCH == 0, "chunk groups");
    constexpr bool MASK_LDS = QT * KT >= 36;          // the large instances: <3, 12> (192 x 192) and <4, *> (256 queries)
    __shared__ __attribute__((aligned(16))) char smem[2 * QR * 128 + X_BYTES + QR * 4 + (MASK_LDS ? KR * 4 : 0)];
    char* sQ = smem; char* sDO = smem + QR * 128; char* sK = sDO + QR * 128; char* sV = sK + KR * 128;
    float* s_lse = reinterpret_cast<float*>(sK + X_BYTES);
    float* s_mask = s_lse + QR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, i15 = lane & 15;
    const int bh = blockIdx.x, head = bh % a.h, b = bh / a.h;
    const float* mb = a.mask + (int64_t)b * a.Skv;
    ATTN_RSTAMP(8);
    ATTN_STAMP(0);
    stage_rows<QR>(sQ, a.Q + (int64_t)b * a.Sq * a.ldq + head * HD, a.ldq, a.Sq, tid);
    stage_rows<QR>(sDO, a.dO + (int64_t)b * a.Sq * a.lddo + head * HD, a.lddo, a.Sq, tid);
    stage_rows<KR>(sK, a.K + (int64_t)b * a.Skv * a.ldk + head * HD, a.ldk, a.Skv, tid);
    stage_rows<KR>(sV, a.V + (int64_t)b * a.Skv * a.ldv + head * HD, a.ldv, a.Skv, tid);
    if (tid < QR) s_lse[tid] = tid < a.Sq ? a.lse[(int64_t)bh * a.Sq + tid] : INFINITY;
    if constexpr (MASK_LDS) {
        static_assert(KR <= 256, "one mask element per thread");
        if (tid < KR) s_mask[tid] = tid < a.Skv ? mb[tid] : -INFINITY;
    }
    ATTN_STAMP(1);
    __syncthreads();
    ATTN_STAMP(2);
    if (ICKA_ATTN_ABLATE == 2) {   // staging + stores only
        for (int i = 0; i < QR * 8 / 256; ++i) {
            const int q = tid + 256 * i, r = q >> 3, c = q & 7;
            if (r < a.Sq) *reinterpret_cast<u32x4*>(a.dQ + ((int64_t)b * a.Sq + r) * a.lddq + head * HD + c * 8) =
                *reinterpret_cast<const u32x4*>(sQ + off_t(r, c)) ^ *reinterpret_cast<const u32x4*>(sDO + off_t(r, c));
        }
        for (int i = 0; i < KR * 8 / 256; ++i) {
            const int q = tid + 256 * i, r = q >> 3, c = q & 7;
            if (r < a.Skv) {
                *reinterpret_cast<u32x4*>(a.dK + ((int64_t)b * a.Skv + r) * a.lddk + head * HD + c * 8) =
                    *reinterpret_cast<const u32x4*>(sK + off_t(r, c));
                *reinterpret_cast<u32x4*>(a.dV + ((int64_t)b * a.Skv + r) * a.lddv + head * HD + c * 8) =
                    *reinterpret_cast<const u32x4*>(sV + off_t(r, c));
            }
        }
        return;
    }

    f32x4 mk[MASK_LDS ? 1 : KT];   // additive mask of key 16*kt + 4*g + r; -inf past the end (large heads: from LDS)
    if constexpr (!MASK_LDS) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = 16 * kt + 4 * g + r;
                const float mv = mb[key < a.Skv ? key : a.Skv - 1];   // clamped load + select: no divergent branch
                mk[kt][r] = key < a.Skv ? mv : -INFINITY;
            }
    }
    // Large heads with dropout: all keep/drop decisions of this lane's (query, key) pairs are hashed FIRST, while nothing
    // else is live, into one bit each (QT*KT*4 bits = 8 registers at 256 x 256); phase A then only tests bits.  With the
    // hashes inside phase A the <4, 16> instance needed ~60 registers more than the 512 a wave can have (308 spilled,
    // 175 us instead of ~90 per launch at B32 x 16 heads).
    constexpr bool DROP_BITS = MASK_LDS && DROP;
    // KB: the forward left its keep decisions (AttnArgs::keepbits): read them instead of hashing again
    constexpr int KBW = (KT * 4 + 31) / 32;             // words of keep bits per (query, lane group) this instance can hold
    const int wpl = keep_wpl(a.Skv);
    uint32_t dbits[DROP_BITS ? (QT * KT * 4 + 31) / 32 : 1];
    if constexpr (DROP_BITS) {
#pragma unroll
        for (int w = 0; w < (QT * KT * 4 + 31) / 32; ++w) dbits[w] = 0u;
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            const int q = 16 * (QT * wave + qi) + i15;
            const uint32_t idx_row = ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;
            const uint32_t hx = (idx_row + 2u * (uint32_t)g) * ICKA_HASH_C0 + a.drop.s0;
            uint32_t kbw[KBW];
            if constexpr (KB) {
                const uint32_t* kp = a.keepbits + (((int64_t)bh * a.Sq + (q < a.Sq ? q : 0)) * 4 + g) * wpl;
#pragma unroll
                for (int w = 0; w < KBW; ++w) kbw[w] = (w < wpl && q < a.Sq) ? kp[w] : 0u;
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int bit = (qi * KT + kt) * 4;
                uint32_t nib;
                if constexpr (KB) nib = (kbw[(kt * 4) >> 5] >> ((kt * 4) & 31)) & 0xFu;
                else nib = drop_nibble_x<true>(a.drop, hx + (uint32_t)(8 * kt) * ICKA_HASH_C0);
                dbits[bit >> 5] |= nib << (bit & 31);
            }
        }
#pragma unroll
        for (int w = 0; w < (QT * KT * 4 + 31) / 32; ++w) asm volatile("" : "+v"(dbits[w]));   // the hashes stay up here
    }
    // ---- phase A: this wave owns queries [16*QT*wave, +16*QT) against every key: P, dP, delta = rowsum(P.dP), dS,
    //      dQ^T = K^T.dS^T straight from the accumulators.  Pd = dropout(P) and dS stay packed in registers for the
    //      exchange below (lane: query i15, keys 16kt + 4g .. +3 -> 8 bytes per kt).
    u32x2 pdp[QT][KT], dsp[QT][KT];
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
        const int q = 16 * (QT * wave + qi) + i15;
        const bf16x8 qf0 = frag_row(sQ, 16 * (QT * wave + qi), 0, lane), qf1 = frag_row(sQ, 16 * (QT * wave + qi), 1, lane);
        const bf16x8 do0 = frag_row(sDO, 16 * (QT * wave + qi), 0, lane), do1 = frag_row(sDO, 16 * (QT * wave + qi), 1, lane);
        const float lse_q = s_lse[q];
        const uint32_t idx_row = ((uint32_t)bh * (uint32_t)a.Sq + (uint32_t)q) * (uint32_t)a.Skv;
        const uint32_t hx = (idx_row + 2u * (uint32_t)g) * ICKA_HASH_C0 + a.drop.s0;   // pair index base (see drop_pair)
        uint32_t kbw[KBW];
        if constexpr (KB && !DROP_BITS) {
            const uint32_t* kp = a.keepbits + (((int64_t)bh * a.Sq + (q < a.Sq ? q : 0)) * 4 + g) * wpl;
#pragma unroll
            for (int w = 0; w < KBW; ++w) kbw[w] = (w < wpl && q < a.Sq) ? kp[w] : 0u;
        }
        // P and dropout-masked dP of this query row block.  Large heads keep P bf16-packed (dS is rounded to bf16 for its
        // MFMA anyway; dP - delta, where the cancellation is, stays f32): 32 registers fewer at the 512-register cap
        f32x4 pr[MASK_LDS ? 1 : KT], dpm[KT];
        float dl = 0.f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 sc = mfma16(frag_row(sK, 16 * kt, 0, lane), qf0, f32x4{0.f, 0.f, 0.f, 0.f});
            sc = mfma16(frag_row(sK, 16 * kt, 1, lane), qf1, sc);
            f32x4 dp = mfma16(frag_row(sV, 16 * kt, 0, lane), do0, f32x4{0.f, 0.f, 0.f, 0.f});
            dp = mfma16(frag_row(sV, 16 * kt, 1, lane), do1, dp);
            f32x4 pd, pvv;
            f32x4 mkv;
            if constexpr (MASK_LDS) mkv = *reinterpret_cast<const f32x4*>(s_mask + 16 * kt + 4 * g);
            else mkv = mk[kt];
            const uint32_t hk = hx + (uint32_t)(8 * kt) * ICKA_HASH_C0;
            // large heads run at the 512-register cap: keep hipcc from overlapping the key tiles of a row block (it hoists
            // the fragment reads, MFMAs and hashes of later tiles above the softmax of this one: +250 live registers)
            if constexpr (MASK_LDS) __builtin_amdgcn_sched_barrier(0);
            float dm4[4];
            if constexpr (!DROP_BITS) {
                if constexpr (KB) {
                    // bit -> all-ones / zero (one signed bit-field extract) -> and with the bits of the scale
                    const uint32_t w = kbw[(kt * 4) >> 5], sb = __float_as_uint(a.drop.scale);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dm4[r] = __uint_as_float(sb & (uint32_t)__builtin_amdgcn_sbfe((int)w, ((kt * 4) & 31) + r, 1));
                } else {
                    drop_pair_x<DROP>(a.drop, hk, dm4[0], dm4[1]);
                    drop_pair_x<DROP>(a.drop, hk + ICKA_HASH_C0, dm4[2], dm4[3]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(sc[r] * a.scale + mkv[r] - lse_q);   // mask -inf (key >= Skv) -> exactly 0
                float dm;
                if constexpr (DROP_BITS) {
                    // (a sign-extending bit-field extract + AND with the bits of 1 / (1 - p) -- two operations instead of shift,
                    //  and, compare, select -- was built in round 5 on the strength of profiles/r04_attn_bwd_stamps.txt: at the
                    //  512-register cap of this instance it cost 305 spilled registers instead of 12 and the launch went from
                    //  95.7 to 153.3 us: profiles/NEGATIVE_RESULTS.md)
                    const int bit = (qi * KT + kt) * 4 + r;
                    dm = ((dbits[bit >> 5] >> (bit & 31)) & 1u) ? a.drop.scale : 0.f;
                } else {
                    dm = dm4[r];
                }
                const float dv = dp[r] * dm;
                pvv[r] = pv;
                dpm[kt][r] = dv;
                pd[r] = pv * dm;
                dl += pv * dv;
            }
            if constexpr (MASK_LDS) {
                // large heads keep ONE packed copy per element: P itself (the dropout factor is applied from the keep bits
                // when Pd is written to the exchange buffer in phase B): a second copy does not fit the register file
                pdp[qi][kt] = pack4(pvv[0], pvv[1], pvv[2], pvv[3]);
            } else {
                pr[kt] = pvv;
                pdp[qi][kt] = pack4(pd[0], pd[1], pd[2], pd[3]);
            }
        }
        dl += __shfl_xor(dl, 16, 64);
        dl += __shfl_xor(dl, 32, 64);
        if (g == 0 && q < a.Sq) a.delta[(int64_t)bh * a.Sq + q] = dl;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x4 pk;
            if constexpr (MASK_LDS) {
                const bf16x4 pb = as_bf16x4(pdp[qi][kt]);
                pk = f32x4{bf2f(pb[0]), bf2f(pb[1]), bf2f(pb[2]), bf2f(pb[3])};
            } else {
                pk = pr[kt];
            }
            const f32x4 d = pk * (dpm[kt] - dl) * a.scale;
            dsp[qi][kt] = pack4(d[0], d[1], d[2], d[3]);
        }
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KT / 2; ++ks)
                acc[dt] = mfma16(frag_tr(sK, dt, ks, lane),
                                 join8(as_bf16x4(dsp[qi][2 * ks]), as_bf16x4(dsp[qi][2 * ks + 1])), acc[dt]);
        }
        if (q < a.Sq) {
            bf16_t* row = a.dQ + ((int64_t)b * a.Sq + q) * a.lddq + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<u32x2*>(row + 16 * dt + 4 * g) = pack4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
        }
    }

    ATTN_STAMP(3);
    // ---- phase B: this wave owns keys [16*KW*wave, +16*KW).  dV^T = dO^T.Pd and dK^T = Q^T.dS reduce over the
    //      queries, which live in other waves' registers: Pd, then dS, go through LDS -- the K/V tiles are dead after
    //      phase A and hold exactly one [QR x KR] bf16 matrix, stored as KR/64 column tiles of the off_t image
    //      ([query row][64 keys]) so the reader takes it with the same transposing fragment read as dO^T / Q^T.
    char* sX = sK;
    f32x4 acc_k[KW][4], acc_v[KW][4];   // tile index gi * NCH + j: 16-key tile 4*NCH*gi + NCH*wave + j of the head
#pragma unroll
    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            acc_k[kw][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_v[kw][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {   // 0: Pd -> dV, 1: dS -> dK
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {    // group of NCH 64-key chunks = 16-key tiles [4*NCH*gi, 4*NCH*(gi+1))
            __syncthreads();                 // everyone is done reading K/V (first round) or the previous group
#pragma unroll
            for (int qi = 0; qi < QT; ++qi)
#pragma unroll
                for (int kl = 0; kl < 4 * NCH; ++kl) {
                    const int kt = 4 * NCH * gi + kl;
                    const int qrow = 16 * (QT * wave + qi) + i15;
                    char* dst = sX + (kl >> 2) * CH_BYTES + off_t(qrow, 2 * (kl & 3) + (g >> 1)) + 8 * (g & 1);
                    u32x2 val = pass == 0 ? pdp[qi][kt] : dsp[qi][kt];
                    if constexpr (DROP_BITS) {
                        if (pass == 0) {   // Pd = P * dropout factor, from the packed P and the keep bits
                            const bf16x4 pb = as_bf16x4(val);
                            float pd4[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int bit = (qi * KT + kt) * 4 + r;
                                pd4[r] = ((dbits[bit >> 5] >> (bit & 31)) & 1u) ? bf2f(pb[r]) * a.drop.scale : 0.f;
                            }
                            val = pack4(pd4[0], pd4[1], pd4[2], pd4[3]);
                        }
                    }
                    *reinterpret_cast<u32x2*>(dst) = val;
                }
            __syncthreads();
            ATTN_STAMP(4 + 2 * pass);
            const char* other = pass == 0 ? sDO : sQ;
#pragma unroll
            for (int ks = 0; ks < 2 * QT; ++ks) {   // 32 queries per MFMA k-slot
                bf16x8 xf[NCH];
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int kk = NCH * wave + j;   // 16-key tile owned by this wave, inside the group
                    xf[j] = frag_tr(sX + (kk >> 2) * CH_BYTES, kk & 3, ks, lane);
                }
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16x8 of = frag_tr(other, dt, ks, lane);
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        if (pass == 0) acc_v[gi * NCH + j][dt] = mfma16(of, xf[j], acc_v[gi * NCH + j][dt]);
                        else acc_k[gi * NCH + j][dt] = mfma16(of, xf[j], acc_k[gi * NCH + j][dt]);
                    }
                }
            }
            ATTN_STAMP(5 + 2 * pass);
        }
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int kw = gi * NCH + j;
            const int key = 16 * (4 * NCH * gi + NCH * wave + j) + i15;
            if (key < a.Skv) {
                bf16_t* krow = a.dK + ((int64_t)b * a.Skv + key) * a.lddk + head * HD;
                bf16_t* vrow = a.dV + ((int64_t)b * a.Skv + key) * a.lddv + head * HD;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    *reinterpret_cast<u32x2*>(krow + 16 * dt + 4 * g) =
                        pack4(acc_k[kw][dt][0], acc_k[kw][dt][1], acc_k[kw][dt][2], acc_k[kw][dt][3]);
                    *reinterpret_cast<u32x2*>(vrow + 16 * dt + 4 * g) =
                        pack4(acc_v[kw][dt][0], acc_v[kw][dt][1], acc_v[kw][dt][2], acc_v[kw][dt][3]);
                }
            }
        }
    ATTN_RSTAMP(9);
}

template <int QT, int KT, bool DROP, bool KB>
static void launch_small2(const AttnArgs& a, int mode, hipStream_t st) {
    if (mode == 1) {
        hipLaunchKernelGGL((attn_bwd_small_kernel<QT, KT, DROP, KB>), dim3(a.B * a.h), dim3(256), 0, st, a);
        return;
    }
    // forward: heads of more than 64 queries run as two 64-query blocks (grid.y = 2, K/V staged twice) when that
    // balances the grid: B*h = 384 whole heads put 2 blocks on half of the 256 CUs and 1 on the rest, 768 half heads
    // put 3 on each
    if constexpr (QT > 2) return;   // (QT = 4 exists for the backward only; the forward takes 64-query blocks)
    else if (QT == 2 && (a.B * a.h) % 256 != 0) {
        if (mode == 2) hipLaunchKernelGGL((attn_fwd_small_kernel<1, KT, DROP, true, KB>), dim3(a.B * a.h, 2), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_small_kernel<1, KT, DROP, false, KB>), dim3(a.B * a.h, 2), dim3(256), 0, st, a);
        return;
    }
    if constexpr (QT <= 2) {
        if (mode == 2) hipLaunchKernelGGL((attn_fwd_small_kernel<QT, KT, DROP, true, KB>), dim3(a.B * a.h), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_fwd_small_kernel<QT, KT, DROP, false, KB>), dim3(a.B * a.h), dim3(256), 0, st, a);
    }
}
template <int QT, int KT>
static void launch_small(const AttnArgs& a, int mode, hipStream_t st) {
    if (a.drop.thr && a.keepbits) launch_small2<QT, KT, true, true>(a, mode, st);
    else if (a.drop.thr) launch_small2<QT, KT, true, false>(a, mode, st);
    else launch_small2<QT, KT, false, false>(a, mode, st);
}
// whole-head path when the head fits (returns false -> caller uses the tiled kernels).
// mode 0: forward, 1: backward, 2: forward with fp8 QK^T / PV
// forward only: up to 256 keys (bert-large, seq 256: BASELINE config c4) and any number of queries as 64-query blocks
// (grid.y) that each stage the head's whole K and V (<= 64 KiB) once and keep their 16 x Skv scores per wave in
// registers: no online-softmax rescaling, one barrier, where the tiled kernel re-stages and re-synchronises per 64 keys
template <int KT>
static void launch_small_fwd_blocks(const AttnArgs& a, hipStream_t st) {
    const dim3 grid(a.B * a.h, (a.Sq + 63) / 64);
    if (a.drop.thr && a.keepbits) hipLaunchKernelGGL((attn_fwd_small_kernel<1, KT, true, false, true>), grid, dim3(256), 0, st, a);
    else if (a.drop.thr) hipLaunchKernelGGL((attn_fwd_small_kernel<1, KT, true, false, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((attn_fwd_small_kernel<1, KT, false, false, false>), grid, dim3(256), 0, st, a);
}
static bool try_small(const AttnArgs& a, int mode, hipStream_t st) {
    if (mode == 0 && a.Skv <= 256 && (a.Sq > 128 || a.Skv > 128)) {
        if (a.Skv > 192) launch_small_fwd_blocks<16>(a, st);
        else if (a.Skv > 128) launch_small_fwd_blocks<12>(a, st);
        else if (a.Skv > 64) launch_small_fwd_blocks<8>(a, st);
        else launch_small_fwd_blocks<4>(a, st);
        return true;
    }
    if (mode == 1 && a.Sq <= 256 && a.Skv <= 256 && (a.Sq > 128 || a.Skv > 128)) {
        // whole heads of up to 256 queries, one block per CU (192 x 192 has its own instance: the prompt-spliced sequences
        // of the published model are 180 long)
        if (a.Sq <= 192 && a.Skv <= 192 && a.Skv > 128) launch_small<3, 12>(a, mode, st);
        else if (a.Skv > 128) launch_small<4, 16>(a, mode, st);
        else if (a.Skv > 64) launch_small<4, 8>(a, mode, st);
        else launch_small<4, 4>(a, mode, st);
        return true;
    }
    if (a.Sq > 128 || a.Skv > 128) return false;
    const bool q2 = a.Sq > 64, k8 = a.Skv > 64;
    if (q2 && k8) launch_small<2, 8>(a, mode, st);
    else if (q2) launch_small<2, 4>(a, mode, st);
    else if (k8) launch_small<1, 8>(a, mode, st);
    else launch_small<1, 4>(a, mode, st);
    return true;
}

inline bool ok16(const void* p, int64_t ld) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld % 8 == 0; }

#ifdef ICKA_ATTN_STAMP
unsigned long long* g_attn_stamp = nullptr;   // diagnostic builds only (tools/attn_stamp.py)
#endif

}  // namespace

#ifdef ICKA_ATTN_STAMP
extern "C" void icka_diag_attn_stamp_buffer(void* p) { g_attn_stamp = (unsigned long long*)p; }
#endif

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
    if (try_small(a, 0, (hipStream_t)stream)) {
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    const int grid = B * heads * ((Sq + TILE - 1) / TILE);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_attn_fwd_fp8(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                 const float* add_mask, void* O, int64_t ldo, float* lse, int32_t B, int32_t heads,
                                 int32_t Sq, int32_t Skv, float scale, float p_drop, uint64_t seed, void* stream) {
    if (!Q || !K || !V || !add_mask || !O) return ICKA_E_ARG;
    if (B <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0 || Sq > 128 || Skv > 128) return ICKA_E_SHAPE;
    if ((int64_t)B * heads * Sq * Skv >= (1ll << 32)) return ICKA_E_SHAPE;
    if (!ok16(Q, ldq) || !ok16(K, ldk) || !ok16(V, ldv) || !ok16(O, ldo)) return ICKA_E_ALIGN;
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
    a.mask = add_mask; a.Ow = (bf16_t*)O; a.ldo = ldo; a.lse = lse;
    a.B = B; a.h = heads; a.Sq = Sq; a.Skv = Skv; a.scale = scale; a.drop = make_drop(p_drop, seed);
    try_small(a, 2, (hipStream_t)stream);
    ICKA_CHECK_LAUNCH();
    return 0;
}

// Forward with an additional fp16 copy of the context ("mixed16": fp16 operand of the out-proj GEMM; the bf16 copy stays
// the operand of the weight-gradient GEMM and of the attention backward).  flags: ICKA_ATTN_FP8 selects the fp8 QK^T / PV
// form, ICKA_ATTN_TILED forces the tiled flash-style kernels for every shape (per call: no process-wide switch).
extern "C" int icka_attn_fwd_ex(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                                const float* add_mask, void* O, void* O_f16, int64_t ldo, float* lse, int32_t B,
                                int32_t heads, int32_t Sq, int32_t Skv, float scale, float p_drop, uint64_t seed,
                                int32_t flags, void* keep_bits, void* stream) {
    if (!Q || !K || !V || !add_mask || !O) return ICKA_E_ARG;
    if (flags & ~(ICKA_ATTN_FP8 | ICKA_ATTN_TILED)) return ICKA_E_ARG;
    const bool fp8 = (flags & ICKA_ATTN_FP8) != 0, tiled = (flags & ICKA_ATTN_TILED) != 0;
    if (fp8 && tiled) return ICKA_E_ARG;    // the fp8 form exists as a whole-head kernel only
    if (B <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0) return ICKA_E_SHAPE;
    if (fp8 && (Sq > 128 || Skv > 128)) return ICKA_E_SHAPE;
    if ((int64_t)B * heads * Sq * Skv >= (1ll << 32)) return ICKA_E_SHAPE;
    if (!ok16(Q, ldq) || !ok16(K, ldk) || !ok16(V, ldv) || !ok16(O, ldo) || (O_f16 && !ok16(O_f16, ldo))) return ICKA_E_ALIGN;
    AttnArgs a{};
    a.Q = (const bf16_t*)Q; a.ldq = ldq; a.K = (const bf16_t*)K; a.ldk = ldk; a.V = (const bf16_t*)V; a.ldv = ldv;
    a.mask = add_mask; a.Ow = (bf16_t*)O; a.Ow16 = (_Float16*)O_f16; a.ldo = ldo; a.lse = lse;
    a.B = B; a.h = heads; a.Sq = Sq; a.Skv = Skv; a.scale = scale; a.drop = make_drop(p_drop, seed);
    a.keepbits = a.drop.thr ? (uint32_t*)keep_bits : nullptr;
    if (fp8) {
        try_small(a, 2, (hipStream_t)stream);
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    if (!tiled && try_small(a, 0, (hipStream_t)stream)) {
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    const int grid = B * heads * ((Sq + TILE - 1) / TILE);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    if (a.keepbits) {   // the tiled forward keeps no bits: leave them for a whole-head backward with one extra launch
        const int64_t rows = (int64_t)B * heads * Sq, n = rows * 4 * keep_wpl(Skv);
        int64_t blocks = (n + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(attn_keepbits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a.keepbits, rows, Skv, a.drop);
        ICKA_CHECK_LAUNCH();
    }
    return 0;
}
extern "C" int64_t icka_attn_keepbits_words(int32_t B, int32_t heads, int32_t Sq, int32_t Skv) {
    if (B <= 0 || heads <= 0 || Sq <= 0 || Skv <= 0) return 0;
    return (int64_t)B * heads * Sq * 4 * ((Skv + 127) >> 7);
}

extern "C" int icka_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                             const float* add_mask, const void* O, int64_t ldo, const void* dO, int64_t lddo,
                             const float* lse, float* delta, void* dQ, int64_t lddq, void* dK, int64_t lddk,
                             void* dV, int64_t lddv, int32_t B, int32_t heads, int32_t Sq, int32_t Skv, float scale,
                             float p_drop, uint64_t seed, const void* keep_bits, int32_t flags, void* stream) {
    if (!Q || !K || !V || !add_mask || !O || !dO || !lse || !delta || !dQ || !dK || !dV) return ICKA_E_ARG;
    if (flags & ~ICKA_ATTN_TILED) return ICKA_E_ARG;
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
    a.keepbits = a.drop.thr ? const_cast<uint32_t*>((const uint32_t*)keep_bits) : nullptr;   // (whole-head kernel only)
    hipStream_t st = (hipStream_t)stream;
#ifdef ICKA_ATTN_STAMP
    a.stamp = g_attn_stamp;
#endif
    if (!(flags & ICKA_ATTN_TILED) && try_small(a, 1, st)) {
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    // the dQ kernel computes delta = rowsum(P . dP) itself (first pass) and leaves it for the dK/dV kernel
    const dim3 gq(B * heads * ((Sq + TILE - 1) / TILE));
    switch ((Skv + TILE - 1) / TILE) {   // <= 256 keys: P and dP stay in registers between the two passes
        case 1: hipLaunchKernelGGL(attn_bwd_dq_keep_kernel<1>, gq, dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(attn_bwd_dq_keep_kernel<2>, gq, dim3(256), 0, st, a); break;
        case 3: hipLaunchKernelGGL(attn_bwd_dq_keep_kernel<3>, gq, dim3(256), 0, st, a); break;
        case 4: hipLaunchKernelGGL(attn_bwd_dq_keep_kernel<4>, gq, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(attn_bwd_dq_kernel, gq, dim3(256), 0, st, a); break;
    }
    ICKA_CHECK_LAUNCH();
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * heads * ((Skv + TILE - 1) / TILE)), dim3(256), 0, st, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}
