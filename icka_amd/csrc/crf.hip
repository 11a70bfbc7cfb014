// Linear-chain CRF on the emissions of the tagging head: log-likelihood, its gradient
// and Viterbi decoding.  The reference calls the third-party package `torchcrf` (pytorch-crf, not vendored, un-pinned:
// Cross_Modal_Interaction_Module.py:3, :911, :1045-1057; my_bert/cl_modeling.py:30, :1269, :1380-1386); the
// semantics here follow that package's published algorithm (v0.7.2: CRF.forward / _compute_score /
// _compute_normalizer / _viterbi_decode), restated for the tests in oracle/crf_oracle.py.
//
// Sequential over the S positions, tiny per step (C <= 64 tags, C x C transitions): ONE WAVE PER SAMPLE, tag j on
// lane j, the running scores and the transition matrix in LDS.  fp32 throughout (log-sum-exp recursions).
//   mask semantics of the package: step 0 is always on; at step t >= 1 the recursion advances only where mask[t] != 0
//   (`torch.where(mask[i], next_score, score)`), the gold-path score adds transitions[tags[t-1], tags[t]] * mask[t],
//   and the end transition is taken at position sum(mask) - 1.
#include <math.h>

#include "common.h"

namespace {

constexpr int CRF_MAX_SC = 12288;   // S * C floats of per-position scores kept in LDS by the gradient kernel (48 KiB)

struct CrfArgs {
    const float* e; int64_t ld_s;   // emissions [B, S, C]: element (b, t, j) at e[(b*S + t)*ld_s + j]
    const int64_t* tags; const int64_t* mask;   // [B, S] (mask may be NULL = all on)
    const float* start; const float* end; const float* trans;   // [C], [C], [C, C] (from, to)
    float* llh;            // [B] log-likelihood of the gold path
    const float* gllh;     // [B] upstream d loss / d llh (gradient kernel)
    float* de; int64_t ld_ds;   // [B, S, C] gradient w.r.t. the emissions
    float* dstart; float* dend; float* dtrans;   // accumulated (+=) with atomics over the samples
    int64_t* best; float* best_score;   // Viterbi: [B, S] tags (-1 past the end), [B]
    int B, S, C;
};

// gold tag, clamped into range (positions that are masked out may carry arbitrary pad labels)
__device__ __forceinline__ int tag_at(const int64_t* tg, int t, int C) {
    const int64_t y = tg[t];
    return y < 0 ? 0 : (y >= C ? C - 1 : (int)y);
}
__device__ __forceinline__ bool on(const CrfArgs& a, int b, int t) {
    return t == 0 || a.mask == nullptr || a.mask[(int64_t)b * a.S + t] != 0;
}

// log-sum-exp (or max) over i of  s_alpha[i] + s_T[i*C + j]  for this lane's tag j
template <bool MAXONLY>
__device__ __forceinline__ float reduce_from(const float* s_alpha, const float* s_T, int C, int j, int* arg) {
    float m = -INFINITY;
    int am = 0;
    for (int i = 0; i < C; ++i) {
        const float v = s_alpha[i] + s_T[i * C + j];
        if (v > m) { m = v; am = i; }
    }
    if (MAXONLY) { *arg = am; return m; }
    float s = 0.f;
    for (int i = 0; i < C; ++i) s += __expf(s_alpha[i] + s_T[i * C + j] - m);
    return m + __logf(s);
}

// gold-path score, lanes strided over the positions
__device__ __forceinline__ float gold_score(const CrfArgs& a, int b, int lane) {
    const int64_t* tg = a.tags + (int64_t)b * a.S;
    float s = 0.f;
    int cnt = 0;
    for (int t = lane; t < a.S; t += 64) {
        const bool act = on(a, b, t);
        cnt += (a.mask == nullptr || a.mask[(int64_t)b * a.S + t] != 0) ? 1 : 0;
        if (!act) continue;
        const int y = tag_at(tg, t, a.C);
        s += a.e[((int64_t)b * a.S + t) * a.ld_s + y];
        s += t == 0 ? a.start[y] : a.trans[tag_at(tg, t - 1, a.C) * a.C + y];
    }
    s = wave_sum(s);
    int len = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);
    const int last = len > 0 ? len - 1 : 0;
    return s + a.end[tag_at(tg, last, a.C)];
}

// ------------------------------------------------------------------------------------------------ log-likelihood
__global__ __launch_bounds__(64) void crf_llh_kernel(const CrfArgs a) {
    __shared__ float s_T[64 * 64];
    __shared__ float s_alpha[64];
    const int b = blockIdx.x, j = threadIdx.x, C = a.C;
    for (int i = j; i < C * C; i += 64) s_T[i] = a.trans[i];
    const float num = gold_score(a, b, j);
    const float* eb = a.e + (int64_t)b * a.S * a.ld_s;
    float alpha = j < C ? a.start[j] + eb[j] : -INFINITY;
    for (int t = 1; t < a.S; ++t) {
        if (!on(a, b, t)) continue;   // wave-uniform
        __syncthreads();   // block = one wave: orders the LDS exchange
        if (j < C) s_alpha[j] = alpha;
        __syncthreads();   // block = one wave: orders the LDS exchange
        if (j < C) alpha = reduce_from<false>(s_alpha, s_T, C, j, nullptr) + eb[(int64_t)t * a.ld_s + j];
    }
    float v = j < C ? alpha + a.end[j] : -INFINITY;
    const float m = wave_max(v);
    const float z = m + __logf(wave_sum(j < C ? __expf(v - m) : 0.f));
    if (j == 0) a.llh[b] = num - z;
}

// ------------------------------------------------------------------------------------------------------ gradient
// d loss / d (emissions, start, end, transitions) = gllh[b] * d llh / d(.) with
//   d llh / d e[t][j] = on(t) * (1[tags[t] = j] - P(y_t = j)),  pairwise marginals for the transitions.
__global__ __launch_bounds__(64) void crf_grad_kernel(const CrfArgs a) {
    __shared__ float s_T[64 * 64];
    __shared__ float s_dT[64 * 64];
    __shared__ float s_vec[64];
    __shared__ float s_al[CRF_MAX_SC];   // alpha[t][j] after step t
    const int b = blockIdx.x, j = threadIdx.x, C = a.C, S = a.S;
    for (int i = j; i < C * C; i += 64) { s_T[i] = a.trans[i]; s_dT[i] = 0.f; }
    const float g = a.gllh[b];
    const float* eb = a.e + (int64_t)b * S * a.ld_s;
    const int64_t* tg = a.tags + (int64_t)b * S;
    float* deb = a.de + (int64_t)b * S * a.ld_ds;

    // forward recursion, every alpha kept
    float alpha = j < C ? a.start[j] + eb[j] : -INFINITY;
    if (j < C) s_al[j] = alpha;
    int len = 1;
    for (int t = 1; t < S; ++t) {
        const bool act = on(a, b, t);
        len += (a.mask == nullptr || a.mask[(int64_t)b * S + t] != 0) ? 1 : 0;
        if (act) {
            __syncthreads();   // block = one wave: orders the LDS exchange
            if (j < C) alpha = reduce_from<false>(s_al + (t - 1) * C, s_T, C, j, nullptr) + eb[(int64_t)t * a.ld_s + j];
        }
        if (j < C) s_al[t * C + j] = alpha;
    }
    if (a.mask != nullptr && a.mask[(int64_t)b * S] == 0) len -= 1;   // (the package rejects this; stay consistent)
    __syncthreads();   // block = one wave: orders the LDS exchange
    float v = j < C ? alpha + a.end[j] : -INFINITY;
    const float m = wave_max(v);
    const float logz = m + __logf(wave_sum(j < C ? __expf(v - m) : 0.f));

    // backward recursion: beta[j] belongs to the latest processed active step
    float beta = j < C ? a.end[j] : -INFINITY;
    float dend = j < C ? -__expf(v - logz) : 0.f;   // - P(y_last = j)
    float dstart = 0.f;
    for (int t = S - 1; t >= 0; --t) {
        const bool act = on(a, b, t);
        if (!act) {
            if (j < C) deb[(int64_t)t * a.ld_ds + j] = 0.f;
            continue;
        }
        const float et = j < C ? eb[(int64_t)t * a.ld_s + j] : 0.f;
        const float marg = j < C ? __expf(s_al[t * C + j] + beta - logz) : 0.f;
        const int y = tag_at(tg, t, C);
        if (j < C) deb[(int64_t)t * a.ld_ds + j] = g * ((j == y ? 1.f : 0.f) - marg);
        if (t == 0) { dstart = (j == y ? 1.f : 0.f) - marg; break; }
        // pairwise marginals with the previous alpha, and the gold transition (literal previous position)
        const float u = et + beta;   // e[t][j] + beta_t[j]
        if (j < C) {
            for (int i = 0; i < C; ++i)
                s_dT[i * C + j] -= __expf(s_al[(t - 1) * C + i] + s_T[i * C + j] + u - logz);
            if (j == y) s_dT[tag_at(tg, t - 1, C) * C + j] += 1.f;
        }
        // beta_{prev}[i] = logsumexp_j (T[i][j] + u[j]): lane i loops over j
        __syncthreads();   // block = one wave: orders the LDS exchange
        if (j < C) s_vec[j] = u;
        __syncthreads();   // block = one wave: orders the LDS exchange
        if (j < C) {
            float mx = -INFINITY;
            for (int k = 0; k < C; ++k) mx = fmaxf(mx, s_T[j * C + k] + s_vec[k]);
            float s = 0.f;
            for (int k = 0; k < C; ++k) s += __expf(s_T[j * C + k] + s_vec[k] - mx);
            beta = mx + __logf(s);
        }
        __syncthreads();   // block = one wave: orders the LDS exchange
    }
    const int last = len > 0 ? len - 1 : 0;
    if (j < C) {
        if (j == tag_at(tg, last, C)) dend += 1.f;
        atomicAdd(a.dstart + j, g * dstart);
        atomicAdd(a.dend + j, g * dend);
    }
    __syncthreads();   // block = one wave: orders the LDS exchange
    for (int i = j; i < C * C; i += 64) atomicAdd(a.dtrans + i, g * s_dT[i]);
}

// ------------------------------------------------------------------------------------------------------- Viterbi
__global__ __launch_bounds__(64) void crf_decode_kernel(const CrfArgs a) {
    __shared__ float s_T[64 * 64];
    __shared__ float s_alpha[64];
    __shared__ unsigned char s_bp[CRF_MAX_SC];   // back-pointers [t][j]
    const int b = blockIdx.x, j = threadIdx.x, C = a.C, S = a.S;
    for (int i = j; i < C * C; i += 64) s_T[i] = a.trans[i];
    const float* eb = a.e + (int64_t)b * S * a.ld_s;
    float score = j < C ? a.start[j] + eb[j] : -INFINITY;
    int len = 1;
    for (int t = 1; t < S; ++t) {
        const bool act = on(a, b, t);
        len += (a.mask == nullptr || a.mask[(int64_t)b * S + t] != 0) ? 1 : 0;
        __syncthreads();   // block = one wave: orders the LDS exchange
        if (j < C) s_alpha[j] = score;
        __syncthreads();   // block = one wave: orders the LDS exchange
        if (j < C) {
            int arg = 0;
            const float nx = reduce_from<true>(s_alpha, s_T, C, j, &arg) + eb[(int64_t)t * a.ld_s + j];
            s_bp[t * C + j] = (unsigned char)arg;   // history is recorded for every step (as the package does)
            if (act) score = nx;
        }
    }
    __syncthreads();   // block = one wave: orders the LDS exchange
    if (j < C) s_alpha[j] = score + a.end[j];
    __syncthreads();   // block = one wave: orders the LDS exchange
    if (j == 0) {
        int bt = 0;
        float bs = s_alpha[0];
        for (int i = 1; i < C; ++i)
            if (s_alpha[i] > bs) { bs = s_alpha[i]; bt = i; }
        if (a.best_score) a.best_score[b] = bs;
        int64_t* out = a.best + (int64_t)b * S;
        const int last = len > 0 ? len - 1 : 0;
        for (int t = last + 1; t < S; ++t) out[t] = -1;
        out[last] = bt;
        for (int t = last; t >= 1; --t) {   // history[:seq_end] reversed
            bt = s_bp[t * C + bt];
            out[t - 1] = bt;
        }
    }
}

// ============================================================================================================
// Register / shuffle variants for C <= 16 tags (the reference has 13 / 15 labels): the transition column (and row) of
// a lane's tag live in registers, the running scores are exchanged with wave shuffles, the mask is a bit set built
// with one ballot per 64 positions and the next step's emissions are prefetched -- no LDS traffic or barrier inside
// the S-step recursions (the LDS versions above spend ~3.5k cycles per step on them).
constexpr int CM = 16;
constexpr int MAXW = 8;   // mask words: S <= 512

struct MaskBits {
    unsigned long long w[MAXW];
    __device__ __forceinline__ bool on(int t) const { return t == 0 || ((w[t >> 6] >> (t & 63)) & 1ull); }
    __device__ __forceinline__ bool raw(int t) const { return (w[t >> 6] >> (t & 63)) & 1ull; }
};
__device__ __forceinline__ MaskBits load_mask(const CrfArgs& a, int b, int lane, int* len) {
    MaskBits mb;
    int n = 0;
#pragma unroll
    for (int k = 0; k < MAXW; ++k) {
        const int t = 64 * k + lane;
        const bool v = t < a.S && (a.mask == nullptr || a.mask[(int64_t)b * a.S + t] != 0);
        mb.w[k] = __ballot(v);
        n += __popcll(mb.w[k]);
    }
    *len = n;
    return mb;
}

// lse_i (x_i + col[i]) with x_i = value of lane i
__device__ __forceinline__ float lse_shfl(float x, const float (&col)[CM]) {
    float v[CM];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < CM; ++i) { v[i] = __shfl(x, i, 64) + col[i]; m = fmaxf(m, v[i]); }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CM; ++i) s += __expf(v[i] - m);
    return m + __logf(s);
}

// log-domain recursions (every step: 16 shuffles + 16 exp + log on the dependent chain): the fallback of the scaled
// linear-domain kernels below for parameters / emissions whose spread underflows f32
__device__ __noinline__ void crf_llh_small_log(const CrfArgs& a, const MaskBits& mb) {
    const int b = blockIdx.x, j = threadIdx.x, C = a.C, S = a.S;
    float Tc[CM];
#pragma unroll
    for (int i = 0; i < CM; ++i) Tc[i] = (j < C && i < C) ? a.trans[i * C + j] : -INFINITY;
    const float num = gold_score(a, b, j);
    const float* eb = a.e + (int64_t)b * S * a.ld_s;
    const bool act_lane = j < C;
    float alpha = act_lane ? a.start[j] + eb[j] : -INFINITY;
    float en = (act_lane && S > 1) ? eb[a.ld_s + j] : 0.f;
    for (int t = 1; t < S; ++t) {
        const float et = en;
        if (t + 1 < S) en = act_lane ? eb[(int64_t)(t + 1) * a.ld_s + j] : 0.f;
        if (!mb.on(t)) continue;
        const float nx = lse_shfl(alpha, Tc) + et;
        alpha = act_lane ? nx : -INFINITY;
    }
    const float v = act_lane ? alpha + a.end[j] : -INFINITY;
    const float m = wave_max(v);
    const float z = m + __logf(wave_sum(act_lane ? __expf(v - m) : 0.f));
    if (j == 0) a.llh[b] = num - z;
}

__device__ __noinline__ void crf_grad_small_log(const CrfArgs& a, const MaskBits& mb, int len, float* s_al) {
    const int b = blockIdx.x, j = threadIdx.x, C = a.C, S = a.S;
    const bool act_lane = j < C;
    float Tc[CM], Tr[CM], dT[CM];   // column j (into tag j), row j (out of tag j), gradient column j
#pragma unroll
    for (int i = 0; i < CM; ++i) {
        Tc[i] = (act_lane && i < C) ? a.trans[i * C + j] : -INFINITY;
        Tr[i] = (act_lane && i < C) ? a.trans[j * C + i] : -INFINITY;
        dT[i] = 0.f;
    }
    const float g = a.gllh[b];
    const float* eb = a.e + (int64_t)b * S * a.ld_s;
    const int64_t* tg = a.tags + (int64_t)b * S;
    float* deb = a.de + (int64_t)b * S * a.ld_ds;

    float alpha = act_lane ? a.start[j] + eb[j] : -INFINITY;
    if (act_lane) s_al[j] = alpha;
    float en = (act_lane && S > 1) ? eb[a.ld_s + j] : 0.f;
    for (int t = 1; t < S; ++t) {
        const float et = en;
        if (t + 1 < S) en = act_lane ? eb[(int64_t)(t + 1) * a.ld_s + j] : 0.f;
        if (mb.on(t)) {
            const float nx = lse_shfl(alpha, Tc) + et;
            alpha = act_lane ? nx : -INFINITY;
        }
        if (act_lane) s_al[t * C + j] = alpha;
    }
    __syncthreads();
    const float v = act_lane ? alpha + a.end[j] : -INFINITY;
    const float m = wave_max(v);
    const float logz = m + __logf(wave_sum(act_lane ? __expf(v - m) : 0.f));

    float beta = act_lane ? a.end[j] : -INFINITY;
    float dend = act_lane ? -__expf(v - logz) : 0.f;
    float dstart = 0.f;
    en = act_lane ? eb[(int64_t)(S - 1) * a.ld_s + j] : 0.f;
    int yn = tag_at(tg, S - 1, C);
    for (int t = S - 1; t >= 0; --t) {
        const float et = en;
        const int y = yn;
        if (t > 0) { en = act_lane ? eb[(int64_t)(t - 1) * a.ld_s + j] : 0.f; yn = tag_at(tg, t - 1, C); }
        if (!mb.on(t)) {
            if (act_lane) deb[(int64_t)t * a.ld_ds + j] = 0.f;
            continue;
        }
        const float marg = act_lane ? __expf(s_al[t * C + j] + beta - logz) : 0.f;
        if (act_lane) deb[(int64_t)t * a.ld_ds + j] = g * ((j == y ? 1.f : 0.f) - marg);
        if (t == 0) { dstart = (j == y ? 1.f : 0.f) - marg; break; }
        const float u = act_lane ? et + beta : -INFINITY;   // e[t][j] + beta_t[j]
        if (act_lane) {
#pragma unroll
            for (int i = 0; i < CM; ++i)
                if (i < C) dT[i] -= __expf(s_al[(t - 1) * C + i] + Tc[i] + u - logz);
        }
        if (j == y) {   // gold transition (literal previous position), yn = tag at t-1
#pragma unroll
            for (int i = 0; i < CM; ++i) dT[i] += (i == yn) ? 1.f : 0.f;
        }
        const float nb = lse_shfl(u, Tr);   // beta_{prev}[j] = lse_k (T[j][k] + u[k])
        beta = act_lane ? nb : -INFINITY;
    }
    const int last = len > 0 ? len - 1 : 0;
    if (act_lane) {
        if (j == tag_at(tg, last, C)) dend += 1.f;
        atomicAdd(a.dstart + j, g * dstart);
        atomicAdd(a.dend + j, g * dend);
#pragma unroll
        for (int i = 0; i < CM; ++i)
            if (i < C) atomicAdd(a.dtrans + i * C + j, g * dT[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Scaled linear-domain forward-backward (C <= 16): the same quantities as the log-domain recursions with NO transcendental
// on the dependent chain -- a_t[j] = (sum_i a_{t-1}[i] E[i][j]) x_t[j] / c_t with E = exp(trans - max trans),
// x_t = exp(e_t - max_j e_t) (computed one step ahead, off the chain) and c_t = sum_i a_{t-1}[i] (the normaliser lags one
// step, so its DPP row sum + reciprocal run beside the dot product), so a step is 16 lane reads + FMAs and a multiply; log Z = sum_t (log c_t + max_j e_t + max trans) + ... accumulates
// beside the chain.  Backward: bhat_{t-1}[i] = sum_j E[i][j] x_t[j] bhat_t[j] / c_t, marginal_t = a_t . bhat_t,
// pair marginals a_{t-1}[i] E[i][j] x_t[j] bhat_t[j] / c_t.  If a normaliser underflows (spreads beyond ~e^80, e.g.
// -1e4 "forbidden" transitions into every reachable tag) or a marginal is not finite the sample is redone by the
// log-domain body above: same results either way up to f32 rounding.
// 16-lane row reductions on the DPP path (row_ror: no LDS crossbar, ~4 VALU ops): every lane of the row gets the result
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f<0x128>(v);   // row_ror:8
    v += dpp_f<0x124>(v);   // row_ror:4
    v += dpp_f<0x122>(v);   // row_ror:2
    v += dpp_f<0x121>(v);   // row_ror:1
    return v;
}
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_f<0x128>(v));
    v = fmaxf(v, dpp_f<0x124>(v));
    v = fmaxf(v, dpp_f<0x122>(v));
    v = fmaxf(v, dpp_f<0x121>(v));
    return v;
}
__device__ __forceinline__ float lane_val(float x, int i) {   // value of lane i (compile-time i) as a scalar operand
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), i));
}
// sum_i (value of lane i) * col[i]
__device__ __forceinline__ float dot_shfl(float x, const float (&col)[CM]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < CM; i += 4) {
        s0 = fmaf(lane_val(x, i), col[i], s0);
        s1 = fmaf(lane_val(x, i + 1), col[i + 1], s1);
        s2 = fmaf(lane_val(x, i + 2), col[i + 2], s2);
        s3 = fmaf(lane_val(x, i + 3), col[i + 3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}
constexpr float CRF_TINY = 1e-30f;
constexpr int CRF_LIN_SC = 4096;   // S * C of the scaled form: x_t and a_t tiles of 16 KiB each

struct CrfLin {
    float logz;      // log partition function
    float zend;      // sum_j a_last[j] exp(end[j] - mEnd)
    float mEnd;
    float a_last;    // this lane's (semi-normalised) alpha after the last step
};
// Stage the sample: emissions -> LDS in one coalesced sweep (the recursions then never wait for HBM or scratch: a per-step
// global load -- or a dynamically indexed mask word, which lives in scratch -- is a whole step of latency), then per
// position (lanes over t) x_t[j] = exp(e_t[j] - max_j e_t[j]) in place, the step-on flags and the clamped gold tags.
// Returns sum over the ON steps t >= 1 of (max_j e_t[j] + mT) + max_j e_0[j]: the static part of log Z.
__device__ __forceinline__ float crf_stage(const CrfArgs& a, float mT, float* s_x, unsigned char* s_on, unsigned char* s_tag) {
    const int b = blockIdx.x, lane = threadIdx.x, C = a.C, S = a.S;
    const float* eb = a.e + (int64_t)b * S * a.ld_s;
    for (int idx = lane; idx < S * C; idx += 64) {
        const int t = idx / C, j = idx - t * C;
        s_x[idx] = eb[(int64_t)t * a.ld_s + j];
    }
    __syncthreads();
    const int64_t* tg = a.tags + (int64_t)b * S;
    float lzs = 0.f;
    for (int t = lane; t < S; t += 64) {
        float m = -INFINITY;
        for (int j = 0; j < C; ++j) m = fmaxf(m, s_x[t * C + j]);
        for (int j = 0; j < C; ++j) s_x[t * C + j] = __expf(s_x[t * C + j] - m);
        const bool on = t == 0 || a.mask == nullptr || a.mask[(int64_t)b * S + t] != 0;
        s_on[t] = on ? 1 : 0;
        if (on) lzs += t == 0 ? m : m + mT;
        if (s_tag) s_tag[t] = (unsigned char)tag_at(tg, t, C);
    }
    __syncthreads();
    return wave_sum(lzs);
}
// Forward over the staged x.  s_al (may be null): a_t[j] for every t; s_c (may be null): the normaliser used AT on-step t
// (= sum of the alpha state before it).  Returns false if a normaliser left the safe range (wave-uniform).
__device__ __forceinline__ bool crf_forward_lin(const CrfArgs& a, const float (&Ec)[CM], float lz_static, const float* s_x,
                                                const unsigned char* s_on, float* s_al, float* s_c, CrfLin& out) {
    const int j = threadIdx.x, C = a.C, S = a.S;
    const bool act_lane = j < C;
    bool ok = true;
    const float sv = act_lane ? a.start[j] : -INFINITY;
    const float mS = max16(sv);
    float al = act_lane ? __expf(sv - mS) * s_x[j] : 0.f;   // a_0 (its sum divides the NEXT step)
    float lz = mS + lz_static;
    if (s_al && act_lane) s_al[j] = al;
    float xn = (act_lane && S > 1) ? s_x[C + j] : 0.f;
    int onn = S > 1 ? s_on[1] : 0;
    for (int t = 1; t < S; ++t) {
        const float x = xn;
        const int on = onn;
        if (t + 1 < S) { xn = act_lane ? s_x[(t + 1) * C + j] : 0.f; onn = s_on[t + 1]; }
        if (on) {
            // a_t = (a_{t-1} . E) x_t / sum(a_{t-1}): the row sum + reciprocal run beside the 16-term dot product
            const float sm = sum16(al);
            ok = ok && sm > CRF_TINY && sm < 1e30f;
            al = dot_shfl(al, Ec) * (x * (1.f / sm));
            lz += __logf(sm);
            if (s_c && j == 0) s_c[t] = sm;
        }
        if (s_al && act_lane) s_al[t * C + j] = al;
    }
    const float ev = act_lane ? a.end[j] : -INFINITY;
    out.mEnd = max16(ev);
    const float w = act_lane ? al * __expf(ev - out.mEnd) : 0.f;
    out.zend = sum16(w);
    ok = ok && out.zend > CRF_TINY && out.zend < 1e30f;
    out.logz = lz + out.mEnd + __logf(out.zend);
    out.a_last = al;
    // (lanes 16..63 carry zeros / NaNs of their own empty rows: only the first 16 lanes vote)
    ok = ok && isfinite(out.logz);
    return (__builtin_amdgcn_ballot_w64(!ok) & 0xffffull) == 0ull;
}
__device__ __forceinline__ float crf_load_E(const CrfArgs& a, int j, float (&Ec)[CM], float (&Er)[CM], bool rows) {
    const int C = a.C;
    float m = -INFINITY;
    for (int k = j; k < C * C; k += 64) m = fmaxf(m, a.trans[k]);
    m = wave_max(m);
#pragma unroll
    for (int i = 0; i < CM; ++i) {
        Ec[i] = (j < C && i < C) ? __expf(a.trans[i * C + j] - m) : 0.f;
        if (rows) Er[i] = (j < C && i < C) ? __expf(a.trans[j * C + i] - m) : 0.f;
    }
    return m;
}

__global__ __launch_bounds__(64) void crf_llh_small_kernel(const CrfArgs a) {
    __shared__ float s_x[CRF_LIN_SC];
    __shared__ unsigned char s_on[64 * MAXW];
    const int b = blockIdx.x, j = threadIdx.x;
    if (a.S * a.C <= CRF_LIN_SC) {
        float Ec[CM], Er[CM];
        const float mT = crf_load_E(a, j, Ec, Er, false);
        const float lzs = crf_stage(a, mT, s_x, s_on, nullptr);
        CrfLin f;
        if (crf_forward_lin(a, Ec, lzs, s_x, s_on, nullptr, nullptr, f)) {
            const float num = gold_score(a, b, j);
            if (j == 0) a.llh[b] = num - f.logz;
            return;
        }
    }
    int len;
    const MaskBits mb = load_mask(a, b, j, &len);
    crf_llh_small_log(a, mb);
}

__global__ __launch_bounds__(64) void crf_grad_small_kernel(const CrfArgs a) {
    __shared__ float s_buf[CRF_MAX_SC];   // scaled form: a_t[j] | x_t[j] (S * C <= 4096 each); log form: alpha[t][j]
    __shared__ float s_c[64 * MAXW];
    __shared__ unsigned char s_tag[64 * MAXW], s_on[64 * MAXW];
    const int b = blockIdx.x, j = threadIdx.x, C = a.C, S = a.S;
    const bool act_lane = j < C;
    bool ok = S * C <= CRF_LIN_SC;
    if (ok) {
        float* s_al = s_buf;
        float* s_x = s_buf + CRF_LIN_SC;
        float Ec[CM], Er[CM];
        const float mT = crf_load_E(a, j, Ec, Er, true);
        const float lzs = crf_stage(a, mT, s_x, s_on, s_tag);
        CrfLin f;
        ok = crf_forward_lin(a, Ec, lzs, s_x, s_on, s_al, s_c, f);
        __syncthreads();
        if (ok) {
            // x_t[j] / S_{t-1} for every on-step, all rows at once (lanes over t): no division on the backward chain
            int cnt = 0;
            for (int t = j; t < S; t += 64) {
                cnt += (a.mask == nullptr || a.mask[(int64_t)b * S + t] != 0) ? 1 : 0;
                if (t >= 1 && s_on[t]) {
                    const float r = 1.f / s_c[t];
                    for (int k = 0; k < C; ++k) s_x[t * C + k] *= r;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
            const int last = cnt > 0 ? cnt - 1 : 0;
            __syncthreads();
            const float g = a.gllh[b];
            float* deb = a.de + (int64_t)b * S * a.ld_ds;
            float dT[CM];
#pragma unroll
            for (int i = 0; i < CM; ++i) dT[i] = 0.f;
            // bhat after the last step: exp(end - mEnd) / zend  (marginal_last = a_last . bhat)
            float bh = act_lane ? __expf(a.end[j] - f.mEnd) / f.zend : 0.f;
            float dend = act_lane ? -(f.a_last * bh) : 0.f;
            float dstart = 0.f;
            // operands of the step, fetched from LDS one step ahead
            int on_n = s_on[S - 1], y_n = s_tag[S - 1], yp_n = S > 1 ? s_tag[S - 2] : 0;
            float al_n = act_lane ? s_al[(S - 1) * C + j] : 0.f, ap_n = (act_lane && S > 1) ? s_al[(S - 2) * C + j] : 0.f;
            float xc_n = act_lane ? s_x[(S - 1) * C + j] : 0.f;
            for (int t = S - 1; t >= 0; --t) {
                const int on = on_n, y = y_n, yp = yp_n;
                const float alt = al_n, ap = ap_n, xc = xc_n;
                if (t > 0) {
                    on_n = s_on[t - 1]; y_n = yp; yp_n = t > 1 ? s_tag[t - 2] : 0;
                    al_n = ap; ap_n = (act_lane && t > 1) ? s_al[(t - 2) * C + j] : 0.f;
                    xc_n = act_lane ? s_x[(t - 1) * C + j] : 0.f;
                }
                if (!on) {
                    if (act_lane) deb[(int64_t)t * a.ld_ds + j] = 0.f;
                    continue;
                }
                const float marg = alt * bh;
                ok = ok && isfinite(marg);
                if (act_lane) deb[(int64_t)t * a.ld_ds + j] = g * ((j == y ? 1.f : 0.f) - marg);
                if (t == 0) { dstart = (j == y ? 1.f : 0.f) - marg; break; }
                // u[j] = x_t[j] bhat_t[j] / S_{t-1};  pair marginal (i, j) = a_{t-1}[i] E[i][j] u[j]
                const float u = xc * bh;
#pragma unroll
                for (int i = 0; i < CM; ++i) dT[i] = fmaf(-lane_val(ap, i) * Ec[i], u, dT[i]);
                if (j == y) {   // gold transition (literal previous position)
#pragma unroll
                    for (int i = 0; i < CM; ++i) dT[i] += (i == yp) ? 1.f : 0.f;
                }
                bh = act_lane ? dot_shfl(u, Er) : 0.f;   // bhat_{t-1}[j] = sum_k E[j][k] u[k]
            }
            ok = (__builtin_amdgcn_ballot_w64(!ok) & 0xffffull) == 0ull;
            if (ok) {
                if (act_lane) {
                    if (j == s_tag[last]) dend += 1.f;
                    atomicAdd(a.dstart + j, g * dstart);
                    atomicAdd(a.dend + j, g * dend);
#pragma unroll
                    for (int i = 0; i < CM; ++i)
                        if (i < C) atomicAdd(a.dtrans + i * C + j, g * dT[i]);
                }
                return;
            }
        }
        __syncthreads();
    }
    int len;
    const MaskBits mb = load_mask(a, b, j, &len);
    crf_grad_small_log(a, mb, len, s_buf);
}

__global__ __launch_bounds__(64) void crf_decode_small_kernel(const CrfArgs a) {
    __shared__ unsigned char s_bp[CRF_MAX_SC];
    __shared__ float s_e[CRF_LIN_SC];
    __shared__ unsigned char s_on[64 * MAXW];
    __shared__ float s_fin[64];
    const int b = blockIdx.x, j = threadIdx.x, C = a.C, S = a.S;
    const bool act_lane = j < C;
    float Tc[CM];
#pragma unroll
    for (int i = 0; i < CM; ++i) Tc[i] = (act_lane && i < C) ? a.trans[i * C + j] : -INFINITY;
    const float* eb = a.e + (int64_t)b * S * a.ld_s;
    // stage emissions and step flags in LDS (as the likelihood kernels: no HBM / scratch access on the S-step chain);
    // larger S * C keep the per-step global loads
    const bool staged = S * C <= CRF_LIN_SC;
    int len = 0;
    for (int t = j; t < S; t += 64) {
        const bool raw = a.mask == nullptr || a.mask[(int64_t)b * S + t] != 0;
        len += raw ? 1 : 0;
        s_on[t] = (t == 0 || raw) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);
    if (staged)
        for (int idx = j; idx < S * C; idx += 64) {
            const int t = idx / C, k = idx - t * C;
            s_e[idx] = eb[(int64_t)t * a.ld_s + k];
        }
    __syncthreads();
    float score = act_lane ? a.start[j] + (staged ? s_e[j] : eb[j]) : -INFINITY;
    float en = (act_lane && S > 1) ? (staged ? s_e[C + j] : eb[a.ld_s + j]) : 0.f;
    int onn = S > 1 ? s_on[1] : 0;
    for (int t = 1; t < S; ++t) {
        const float et = en;
        const int on = onn;
        if (t + 1 < S) {
            en = act_lane ? (staged ? s_e[(t + 1) * C + j] : eb[(int64_t)(t + 1) * a.ld_s + j]) : 0.f;
            onn = s_on[t + 1];
        }
        float m = -INFINITY;
        int am = 0;
#pragma unroll
        for (int i = 0; i < CM; ++i) {
            const float v = lane_val(score, i) + Tc[i];
            if (v > m) { m = v; am = i; }
        }
        if (act_lane) {
            s_bp[t * C + j] = (unsigned char)am;
            if (on) score = m + et;
        }
    }
    s_fin[j] = act_lane ? score + a.end[j] : -INFINITY;
    __syncthreads();
    if (j == 0) {
        int bt = 0;
        float bs = s_fin[0];
        for (int i = 1; i < C; ++i)
            if (s_fin[i] > bs) { bs = s_fin[i]; bt = i; }
        if (a.best_score) a.best_score[b] = bs;
        int64_t* out = a.best + (int64_t)b * S;
        const int last = len > 0 ? len - 1 : 0;
        for (int t = last + 1; t < S; ++t) out[t] = -1;
        out[last] = bt;
        for (int t = last; t >= 1; --t) {
            bt = s_bp[t * C + bt];
            out[t - 1] = bt;
        }
    }
}

inline bool crf_small(int S, int C) { return C <= CM && S <= 64 * MAXW; }

inline int crf_check(const void* e, const void* start, const void* end, const void* trans, int B, int S, int C) {
    if (!e || !start || !end || !trans) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || C <= 0 || C > 64) return ICKA_E_SHAPE;
    return 0;
}

}  // namespace

extern "C" int icka_crf_llh(const float* emissions, int64_t ld, const int64_t* tags, const int64_t* mask,
                            const float* start, const float* end, const float* trans, float* llh, int32_t B,
                            int32_t S, int32_t C, void* stream) {
    if (int rc = crf_check(emissions, start, end, trans, B, S, C)) return rc;
    if (!tags || !llh || ld < C) return ICKA_E_ARG;
    CrfArgs a{};
    a.e = emissions; a.ld_s = ld; a.tags = tags; a.mask = mask; a.start = start; a.end = end; a.trans = trans;
    a.llh = llh; a.B = B; a.S = S; a.C = C;
    if (crf_small(S, C)) hipLaunchKernelGGL(crf_llh_small_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(crf_llh_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_crf_grad(const float* emissions, int64_t ld, const int64_t* tags, const int64_t* mask,
                             const float* start, const float* end, const float* trans, const float* gllh,
                             float* d_emissions, int64_t ldd, float* d_start, float* d_end, float* d_trans, int32_t B,
                             int32_t S, int32_t C, void* stream) {
    if (int rc = crf_check(emissions, start, end, trans, B, S, C)) return rc;
    if (!tags || !gllh || !d_emissions || !d_start || !d_end || !d_trans || ld < C || ldd < C) return ICKA_E_ARG;
    if ((int64_t)S * C > CRF_MAX_SC) return ICKA_E_SHAPE;
    CrfArgs a{};
    a.e = emissions; a.ld_s = ld; a.tags = tags; a.mask = mask; a.start = start; a.end = end; a.trans = trans;
    a.gllh = gllh; a.de = d_emissions; a.ld_ds = ldd; a.dstart = d_start; a.dend = d_end; a.dtrans = d_trans;
    a.B = B; a.S = S; a.C = C;
    if (crf_small(S, C)) hipLaunchKernelGGL(crf_grad_small_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(crf_grad_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_crf_decode(const float* emissions, int64_t ld, const int64_t* mask, const float* start,
                               const float* end, const float* trans, int64_t* best_tags, float* best_score, int32_t B,
                               int32_t S, int32_t C, void* stream) {
    if (int rc = crf_check(emissions, start, end, trans, B, S, C)) return rc;
    if (!best_tags || ld < C) return ICKA_E_ARG;
    if ((int64_t)S * C > CRF_MAX_SC) return ICKA_E_SHAPE;
    CrfArgs a{};
    a.e = emissions; a.ld_s = ld; a.mask = mask; a.start = start; a.end = end; a.trans = trans;
    a.best = best_tags; a.best_score = best_score; a.B = B; a.S = S; a.C = C;
    if (crf_small(S, C)) hipLaunchKernelGGL(crf_decode_small_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(crf_decode_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}
