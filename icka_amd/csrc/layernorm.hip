// Fused bias + dropout + residual + LayerNorm (forward / backward) and the BertEmbeddings gather + LayerNorm +
// dropout (forward / backward) for gfx950.  HBM/L2-bound row kernels: one 64-lane wave per row, every lane owns
// 16-byte chunks (8 bf16) of the row -> coalesced 1-KiB wave accesses, statistics by wave-level shuffles in fp32.
// Column reductions of the backward (dgamma, dbeta, dbias, token-type rows) are kept in per-lane registers over the
// rows a wave visits, combined per block through LDS and written as per-block partial slabs that a tiny finalize
// kernel sums: no global float atomics on hot addresses, bitwise reproducible.
#include "common.h"
#include "ln_row.h"
#include <cstdlib>

namespace {

constexpr int BWD_BLOCKS = 1024;    // partial slabs per backward launch (one row per wave at M = 4096: 16 waves/CU)
constexpr int SLOTS = 4;            // column-sum slots per slab

template <int NCH>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LnFwdArgs a_) {
    LnFwdArgs a = a_;
    a.drop = drop_resolve(a.drop);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    ln_fwd_rows<NCH>(a, wid, nw, a.M, lane);      // (ln_row.h: the row body shared with gemm.hip's gemm_ln_kernel)
}

struct LnBwdArgs {
    const bf16_t* dy; int64_t lddy; const bf16_t* dy2; int64_t lddy2; const bf16_t* xhat; const float* rstd;
    const float* gamma; bf16_t* dres; int64_t lddres; bf16_t* dx; int64_t lddx; float* partials;
    int M, H; DropCfg drop;
};

// Combine the 4 waves' per-lane column sums through LDS and write this block's slab: partials[blk][slot][H].
template <int NCH, int NS>
__device__ __forceinline__ void flush_columns(float (&acc)[NS][NCH][8], float* lds, float* partials, int H) {
    const int lane = threadIdx.x & 63, nchunk = H >> 3;
    for (int i = threadIdx.x; i < NS * H; i += 256) lds[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) atomicAdd(&lds[s * H + c * 8 + e], acc[s][i][e]);
            }
        }
    __syncthreads();
    float* slab = partials + (int64_t)blockIdx.x * SLOTS * H;
    for (int i = threadIdx.x; i < NS * H; i += 256) slab[i] = lds[i];
}

// Row part of the backward: dres / dx only (pure streaming, one row per wave).  The column reductions (dgamma, dbeta,
// dbias) are a separate pass (ln_cols_kernel): fusing them here forced either few waves (latency-bound rows) or a
// per-block flush of 3*H partial sums that cost more than the rows themselves (measured 28 -> 56 us at 1024 blocks).
// raw (packed bf16) operands of one backward row: issued one row ahead of their use when a wave owns several rows
template <int NCH>
__device__ __forceinline__ void ln_bwd_load_raw(const LnBwdArgs& a, int row, int lane, int nchunk, u32x4 (&dy)[NCH], u32x4 (&dy2)[NCH],
                                                u32x4 (&xh)[NCH]) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            dy[i] = *reinterpret_cast<const u32x4*>(a.dy + (int64_t)row * a.lddy + c * 8);
            if (a.dy2) dy2[i] = *reinterpret_cast<const u32x4*>(a.dy2 + (int64_t)row * a.lddy2 + c * 8);
            xh[i] = *reinterpret_cast<const u32x4*>(a.xhat + (int64_t)row * a.H + c * 8);
        }
    }
}
__device__ __forceinline__ void unpack8(const u32x4 raw, float (&o)[8]) {
    const bf16x8 v = as_bf16x8(raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(v[e]);
}

template <int NCH>
__device__ __forceinline__ void ln_rows_body(const LnBwdArgs& a_, int blk, int nblk) {
    LnBwdArgs a = a_;
    a.drop = drop_resolve(a.drop);
    const int lane = threadIdx.x & 63;
    const int wid = blk * 4 + (threadIdx.x >> 6), nw = nblk * 4;
    const int nchunk = a.H >> 3;
    const float inv_h = 1.f / (float)a.H;
    u32x4 rdy[NCH], rdy2[NCH], rxh[NCH];
    if (wid < a.M) ln_bwd_load_raw<NCH>(a, wid, lane, nchunk, rdy, rdy2, rxh);
    for (int row = wid; row < a.M; row += nw) {
        float xh[NCH][8], gd[NCH][8];
        float s1 = 0.f, s2 = 0.f;
        // the keep decisions of the row's dropout mask FIRST (one bit each), while the loads are in flight: left next to
        // their use after the row reductions, the 8 hashes per chunk sat on the critical path of a latency-bound kernel
        // (+1.1 us per launch at 4096 x 768, tools/ln_bench.py)
        uint32_t kb[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            kb[i] = 0xffu;
            const int c = lane + 64 * i;
            if (a.drop.thr && c < nchunk) {
                const uint32_t base = (uint32_t)row * (uint32_t)a.H + c * 8;
                uint32_t bits = 0u;
#pragma unroll
                for (int e = 0; e < 8; ++e) bits |= (icka_hash(a.drop.s0, a.drop.s1, base + e) >= a.drop.thr ? 1u : 0u) << e;
                kb[i] = bits;
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i) asm volatile("" : "+v"(kb[i]));   // (keep the hashes up here)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float dy[8];
                unpack8(rdy[i], dy);
                if (a.dy2) {
                    float t[8];
                    unpack8(rdy2[i], t);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dy[e] += t[e];
                }
                unpack8(rxh[i], xh[i]);
                float g[8];
                load8f(a.gamma + c * 8, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    gd[i][e] = g[e] * dy[e];
                    s1 += gd[i][e];
                    s2 += gd[i][e] * xh[i][e];
                }
            }
        }
        // a wave that owns another row issues its loads now: they are in flight under the reductions and stores below
        const int nrow = row + nw;
        if (nrow < a.M) ln_bwd_load_raw<NCH>(a, nrow, lane, nchunk, rdy, rdy2, rxh);
        const float c1 = wave_sum(s1) * inv_h, c2 = wave_sum(s2) * inv_h;
        const float rstd = a.rstd[row];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float ds[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ds[e] = rstd * (gd[i][e] - c1 - xh[i][e] * c2);
                if (a.dres) store8(a.dres + (int64_t)row * a.lddres + c * 8, ds);
                if (a.drop.thr) {  // gradient of the dense output: through the (re-generated) dropout mask
#pragma unroll
                    for (int e = 0; e < 8; ++e) ds[e] *= ((kb[i] >> e) & 1u) ? a.drop.scale : 0.f;
                }
                if (a.dx) store8(a.dx + (int64_t)row * a.lddx + c * 8, ds);
            }
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const LnBwdArgs a) {
    ln_rows_body<NCH>(a, blockIdx.x, gridDim.x);
}

// Column part: slab[g][0] = sum_rows dy*xhat, slab[g][1] = sum_rows dy, slab[g][2] = sum_rows dx (if dx), over the
// rows of group g.  Block = 32 chunk-lanes (256 columns) x 8 row-lanes; 16-byte loads, consecutive lanes on
// consecutive chunks.
constexpr int COL_GROUPS = 128;
__device__ __forceinline__ void ln_cols_body(const bf16_t* __restrict__ dy, int64_t lddy,
                                             const bf16_t* __restrict__ dy2, int64_t lddy2,
                                             const bf16_t* __restrict__ xhat, const bf16_t* __restrict__ dx,
                                             int64_t lddx, float* __restrict__ partials, int M, int H,
                                             int rows_per_group, int bx, int by) {
    __shared__ float red[3][8][32][8];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = (bx * 32 + cx) * 8;
    const int r0 = by * rows_per_group;
    int r1 = r0 + rows_per_group; r1 = r1 > M ? M : r1;
    float ag[8], ab[8], ax[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[e] = 0.f; ab[e] = 0.f; ax[e] = 0.f; }
    if (col < H) {
#pragma unroll 4
        for (int r = r0 + ry; r < r1; r += 8) {   // unrolled: 4 rows of independent 16-byte loads in flight per thread
            float d[8], x[8];
            load8(dy + (int64_t)r * lddy + col, d);
            if (dy2) {
                float t[8];
                load8(dy2 + (int64_t)r * lddy2 + col, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) d[e] += t[e];
            }
            load8(xhat + (int64_t)r * H + col, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ag[e] += d[e] * x[e]; ab[e] += d[e]; }
            if (dx) {
                load8(dx + (int64_t)r * lddx + col, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) ax[e] += x[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][ry][cx][e] = ag[e]; red[1][ry][cx][e] = ab[e]; red[2][ry][cx][e] = ax[e]; }
    __syncthreads();
    // 3 slots x 256 columns = 768 sums per block, 3 per thread
    for (int i = threadIdx.x; i < 3 * 256; i += 256) {
        const int slot = i >> 8, cc = i & 255, c = bx * 256 + cc;
        if (c < H) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[slot][k][cc >> 3][cc & 7];
            partials[((int64_t)by * SLOTS + slot) * H + c] = t;
        }
    }
}
__global__ __launch_bounds__(256) void ln_cols_kernel(const bf16_t* __restrict__ dy, int64_t lddy,
                                                      const bf16_t* __restrict__ dy2, int64_t lddy2,
                                                      const bf16_t* __restrict__ xhat, const bf16_t* __restrict__ dx,
                                                      int64_t lddx, float* __restrict__ partials, int M, int H,
                                                      int rows_per_group) {
    ln_cols_body(dy, lddy, dy2, lddy2, xhat, dx, lddx, partials, M, H, rows_per_group, blockIdx.x, blockIdx.y);
}

// Rows and columns of the LayerNorm backward in ONE launch (no dbias: the dense bias gradient rides on the
// weight-gradient GEMM): blocks [0, nx*groups) reduce columns into slabs, the rest stream rows.  The two halves are
// independent, so they share the machine instead of queueing behind each other; the slabs are summed by the finalize
// launch.  (A "last block sums the slabs" variant was measured and dropped: its device-scope release fences write the
// whole L2 back on a multi-XCD part -- 23-36 us per call against 20 us for three separate launches.)
struct LnColsArgs {
    int nx, groups, rows_per_group;
};
template <int NCH>
__global__ __launch_bounds__(256) void ln_bwd_fused_kernel(const LnBwdArgs a, const LnColsArgs c) {
    const int ncol = c.nx * c.groups;
    if ((int)blockIdx.x >= ncol) {
        ln_rows_body<NCH>(a, blockIdx.x - ncol, gridDim.x - ncol);
        return;
    }
    ln_cols_body(a.dy, a.lddy, a.dy2, a.lddy2, a.xhat, nullptr, 0, a.partials, a.M, a.H, c.rows_per_group,
                 blockIdx.x % c.nx, blockIdx.x / c.nx);
}

// out[slot][c] (+)= sum over slabs.  Block = 64 columns x 16 slab-lanes: the slab loop is split 16 ways and combined
// through LDS, so the reduction is a few dependent loads deep instead of nslab.
__global__ __launch_bounds__(1024) void finalize_kernel(const float* __restrict__ partials, int nslab, int H, float* o0,
                                                        float* o1, float* o2, float* o3, int accumulate) {
    __shared__ float red[16][64];
    const int cx = threadIdx.x & 63, sy = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cx;
    float s = 0.f;
    if (idx < SLOTS * H) {
        const int slot = idx / H, c = idx - slot * H;
        for (int b = sy; b < nslab; b += 16) s += partials[((int64_t)b * SLOTS + slot) * H + c];
    }
    red[sy][cx] = s;
    __syncthreads();
    if (sy == 0 && idx < SLOTS * H) {
        const int slot = idx / H, c = idx - slot * H;
        float* out = slot == 0 ? o0 : slot == 1 ? o1 : slot == 2 ? o2 : o3;
        if (out) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][cx];
            out[c] = accumulate ? out[c] + t : t;
        }
    }
}

// ------------------------------------------------------------------------------------------------- embeddings
struct EmbFwdArgs {
    const int64_t* ids; const int64_t* tt; const float* word; const float* pos; const float* type;
    const float* gamma; const float* beta; bf16_t* y; void* yf; bf16_t* xhat; float* rstd;
    int M, S, H, vocab, n_type; float eps; DropCfg drop;
    // prompt splice (icka_embed_prompt_fwd): output position t takes token src[t] of the S_in-long id row, or, for
    // src[t] < 0, prompt vector -1-src[t] of this sample's [P,H] bf16 prompt block; position row = t + pos_offset
    const int32_t* src; const bf16_t* prompt; int S_in, P, pos_offset;
    int yf_f16;   // twin output is fp16 ("mixed16") instead of f32
};

template <int NCH>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbFwdArgs a_) {
    EmbFwdArgs a = a_;
    a.drop = drop_resolve(a.drop);
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nchunk = a.H >> 3;
    const float inv_h = 1.f / (float)a.H;
    for (int row = wid; row < a.M; row += nw) {
        const int sp = row % a.S;
        const int sidx = a.src ? a.src[sp] : sp;
        const bf16_t* prow = sidx < 0 ? a.prompt + ((int64_t)(row / a.S) * a.P + (-1 - sidx)) * a.H : nullptr;
        int64_t id = sidx < 0 ? 0 : a.ids[a.src ? (int64_t)(row / a.S) * a.S_in + sidx : row];
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
        int64_t t = a.tt ? a.tt[row] : 0;
        t = t < 0 ? 0 : (t >= a.n_type ? a.n_type - 1 : t);
        float s[NCH][8];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float w[8], p[8], ty[8];
                if (prow) load8(prow + c * 8, w);
                else load8f(a.word + id * a.H + c * 8, w);
                load8f(a.pos + (int64_t)(sp + a.pos_offset) * a.H + c * 8, p);
                load8f(a.type + t * a.H + c * 8, ty);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[i][e] = (w[e] + p[e]) + ty[e]; sum += s[i][e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) s[i][e] = 0.f;
            }
        }
        const float mean = wave_sum(sum) * inv_h;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            if (lane + 64 * i < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = s[i][e] - mean; sq += d * d; }
            }
        const float rstd = 1.f / sqrtf(wave_sum(sq) * inv_h + a.eps);
        if (lane == 0 && a.rstd) a.rstd[row] = rstd;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float g[8], b[8], xh[8], o[8];
                load8f(a.gamma + c * 8, g);
                load8f(a.beta + c * 8, b);
                const uint32_t base = (uint32_t)row * (uint32_t)a.H + c * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = (s[i][e] - mean) * rstd;
                    o[e] = (g[e] * xh[e] + b[e]) * drop_mul(a.drop, base + e);
                }
                store8(a.y + (int64_t)row * a.H + c * 8, o);
                if (a.yf) store8t(a.yf, a.yf_f16, (int64_t)row * a.H + c * 8, o);
                if (a.xhat) store8(a.xhat + (int64_t)row * a.H + c * 8, xh);
            }
        }
    }
}

struct EmbBwdArgs {
    const bf16_t* dy; const int64_t* ids; const int64_t* tt; const bf16_t* xhat; const float* rstd;
    const float* gamma; float* dword; float* dpos; float* dtype; float* partials;
    int M, S, H, vocab, n_type, padding_idx; DropCfg drop;
    const int32_t* src; bf16_t* dprompt; int S_in, P, pos_offset;   // prompt splice, see EmbFwdArgs (pos kernel only)
    float* dtok;   // row-sparse data-parallel exchange (icka_embed_bwd_rows): the per-token word-gradient row is WRITTEN to
                   // dtok[row][H] (zeros for the padding id) instead of being added into dword
};

template <int NCH>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const EmbBwdArgs a_) {
    EmbBwdArgs a = a_;
    a.drop = drop_resolve(a.drop);
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nchunk = a.H >> 3;
    const float inv_h = 1.f / (float)a.H;
    float acc[4][NCH][8];  // dgamma, dbeta, type row 0, type row 1
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[s][i][e] = 0.f;

    for (int row = wid; row < a.M; row += nw) {
        int64_t id = a.ids[row];
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
        int64_t t = a.tt ? a.tt[row] : 0;
        t = t < 0 ? 0 : (t >= a.n_type ? a.n_type - 1 : t);
        const int sp = row % a.S;
        float xh[NCH][8], gd[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float dy[8], g[8];
                load8(a.dy + (int64_t)row * a.H + c * 8, dy);
                load8(a.xhat + (int64_t)row * a.H + c * 8, xh[i]);
                load8f(a.gamma + c * 8, g);
                const uint32_t base = (uint32_t)row * (uint32_t)a.H + c * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dl = dy[e] * drop_mul(a.drop, base + e);  // gradient at the LayerNorm output
                    gd[i][e] = g[e] * dl;
                    s1 += gd[i][e];
                    s2 += gd[i][e] * xh[i][e];
                    acc[0][i][e] += dl * xh[i][e];
                    acc[1][i][e] += dl;
                }
            }
        }
        const float c1 = wave_sum(s1) * inv_h, c2 = wave_sum(s2) * inv_h;
        const float rstd = a.rstd[row];
        // The row of table gradients goes through LDS so that the global f32 atomics are issued lane-strided: one
        // wave-instruction adds 64 CONSECUTIVE floats (256 contiguous bytes), the full-rate shape of
        // global_atomic_add_f32 (MI355X_MICROARCH "Global float atomics"); 8-per-lane chunks would scatter 32-B pieces.
        float* rowbuf = lds_f + 4 * a.H + (threadIdx.x >> 6) * a.H;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ds = rstd * (gd[i][e] - c1 - xh[i][e] * c2);
                    rowbuf[c * 8 + e] = ds;
                    if (a.n_type <= 2) {
                        acc[2][i][e] += t == 0 ? ds : 0.f;
                        acc[3][i][e] += t == 1 ? ds : 0.f;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < a.H; j += 64) {
            const float ds = rowbuf[j];
            if (a.dtok) a.dtok[(int64_t)row * a.H + j] = id != a.padding_idx ? ds : 0.f;
            else if (id != a.padding_idx) atomicAdd(a.dword + id * a.H + j, ds);
            atomicAdd(a.dpos + (int64_t)sp * a.H + j, ds);
            if (a.n_type > 2) atomicAdd(a.dtype + t * a.H + j, ds);
        }
        __builtin_amdgcn_wave_barrier();
    }
    flush_columns<NCH, 4>(acc, lds_f, a.partials, a.H);
}

// Position-major form of the same backward (the default): ONE BLOCK PER SEQUENCE POSITION s, its 8 waves walk the
// batch.  The position-table gradient of row s is then a plain block-local sum (no atomics, no 32-way same-address
// contention), and the dgamma / dbeta / token-type column sums flush S slabs instead of 1024; only the word-table rows
// still take f32 atomics.  (One row per wave over 1024 blocks measured 77 us at c2.)
template <int NCH>
__global__ __launch_bounds__(512) void embed_bwd_pos_kernel(const EmbBwdArgs a_) {
    EmbBwdArgs a = a_;
    a.drop = drop_resolve(a.drop);
    extern __shared__ __attribute__((aligned(16))) float lds_f[];   // [8 waves][H]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sp = blockIdx.x, B = a.M / a.S;
    const int nchunk = a.H >> 3;
    const float inv_h = 1.f / (float)a.H;
    float acc[5][NCH][8];  // dgamma, dbeta, type row 0, type row 1, position row sp
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[s][i][e] = 0.f;
    float* rowbuf = lds_f + wave * a.H;

    const int sidx = a.src ? a.src[sp] : sp;
    for (int b = wave; b < B; b += 8) {
        const int row = b * a.S + sp;
        int64_t id = sidx < 0 ? 0 : a.ids[a.src ? (int64_t)b * a.S_in + sidx : row];
        id = id < 0 ? 0 : (id >= a.vocab ? a.vocab - 1 : id);
        int64_t t = a.tt ? a.tt[row] : 0;
        t = t < 0 ? 0 : (t >= a.n_type ? a.n_type - 1 : t);
        float xh[NCH][8], gd[NCH][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
                float dy[8], g[8];
                load8(a.dy + (int64_t)row * a.H + c * 8, dy);
                load8(a.xhat + (int64_t)row * a.H + c * 8, xh[i]);
                load8f(a.gamma + c * 8, g);
                const uint32_t base = (uint32_t)row * (uint32_t)a.H + c * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float dl = dy[e] * drop_mul(a.drop, base + e);
                    gd[i][e] = g[e] * dl;
                    s1 += gd[i][e];
                    s2 += gd[i][e] * xh[i][e];
                    acc[0][i][e] += dl * xh[i][e];
                    acc[1][i][e] += dl;
                }
            }
        }
        const float c1 = wave_sum(s1) * inv_h, c2 = wave_sum(s2) * inv_h;
        const float rstd = a.rstd[row];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float ds = rstd * (gd[i][e] - c1 - xh[i][e] * c2);
                    rowbuf[c * 8 + e] = ds;
                    acc[4][i][e] += ds;
                    acc[2][i][e] += t == 0 ? ds : 0.f;
                    acc[3][i][e] += t == 1 ? ds : 0.f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // word-table row: lane-strided f32 atomics (64 consecutive floats per wave instruction)
        if (sidx < 0) {   // prompt position: the row IS the gradient of this sample's prompt vector
            bf16_t* dp = a.dprompt + ((int64_t)b * a.P + (-1 - sidx)) * a.H;
            for (int j = lane; j < a.H; j += 64) dp[j] = f2bf(rowbuf[j]);
        } else if (a.dtok) {
            for (int j = lane; j < a.H; j += 64) {
                const float ds = rowbuf[j];
                a.dtok[(int64_t)row * a.H + j] = id != a.padding_idx ? ds : 0.f;
                if (a.n_type > 2) atomicAdd(a.dtype + t * a.H + j, ds);
            }
        } else if (id != a.padding_idx || a.n_type > 2) {
            for (int j = lane; j < a.H; j += 64) {
                const float ds = rowbuf[j];
                if (id != a.padding_idx) atomicAdd(a.dword + id * a.H + j, ds);
                if (a.n_type > 2) atomicAdd(a.dtype + t * a.H + j, ds);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    // block reduction of the 5 column sums over the 8 waves, one slot at a time through the [8][H] buffer
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) rowbuf[c * 8 + e] = acc[s][i][e];
            }
        }
        __syncthreads();
        for (int j = tid; j < a.H; j += 512) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += lds_f[w * a.H + j];
            if (s < 4) a.partials[((int64_t)sp * SLOTS + s) * a.H + j] = v;
            else a.dpos[(int64_t)(sp + a.pos_offset) * a.H + j] += v;   // this block owns its position row (caller zeroes / accumulates)
        }
    }
}

inline int pick_nch(int H) { return (H / 8 + 63) / 64; }
// rows a forward wave owns (> 1: its next row's loads overlap its stores): two rows per wave from 4096 rows on (c2: 4.534 ->
// 4.489 ms per step, same box, twice: profiles/r04_ln_rows_per_wave.txt; three and more rows per wave are slower again, and the
// backward rows do not gain).  ICKA_TUNE_LN_ROWS_PER_WAVE (1 .. 16; read ONCE at load: same-box A/B runs, tools/ln_bench.py
// starts one process per setting) replaces the automatic choice -- there is no run-time setter: no mutable process state
inline int rows_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 16 ? v : dflt;
}
const int kLnRowsEnv = rows_knob("ICKA_TUNE_LN_ROWS_PER_WAVE", 0);
const int kLnBwdRowsEnv = rows_knob("ICKA_TUNE_LN_BWD_ROWS_PER_WAVE", 1);
inline int row_grid(int M) { int g = (M + 3) / 4; return g > 2048 ? 2048 : (g < 1 ? 1 : g); }
inline int fwd_grid(int M) {
    const int r = kLnRowsEnv ? kLnRowsEnv : (M >= 4096 ? 2 : 1);
    int g = (M + 4 * r - 1) / (4 * r);
    return g > 2048 ? 2048 : (g < 1 ? 1 : g);
}
inline int bwd_grid(int M) { int g = (M + 3) / 4; return g > BWD_BLOCKS ? BWD_BLOCKS : (g < 1 ? 1 : g); }
// row blocks of the fused backward launch (rows per wave: ICKA_TUNE_LN_BWD_ROWS_PER_WAVE, default 1)
inline int bwd_row_grid(int M) {
    const int r = kLnBwdRowsEnv;
    int g = (M + 4 * r - 1) / (4 * r);
    return g > 2048 ? 2048 : (g < 1 ? 1 : g);
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define DISPATCH_NCH(nch, KERNEL, grid, shmem, st, args)                                                 \
    switch (nch) {                                                                                       \
        case 1: hipLaunchKernelGGL((KERNEL<1>), dim3(grid), dim3(256), shmem, st, args); break;          \
        case 2: hipLaunchKernelGGL((KERNEL<2>), dim3(grid), dim3(256), shmem, st, args); break;          \
        case 3: hipLaunchKernelGGL((KERNEL<3>), dim3(grid), dim3(256), shmem, st, args); break;          \
        default: hipLaunchKernelGGL((KERNEL<4>), dim3(grid), dim3(256), shmem, st, args); break;         \
    }

}  // namespace

extern "C" int64_t icka_ln_bwd_workspace_floats(int32_t H) { return (int64_t)BWD_BLOCKS * SLOTS * H; }

static int ln_fwd_impl(int32_t twin_f16, const void* x, int64_t ldx, int32_t x_is_f32, const float* bias, const void* residual,
                           int64_t ldr, int32_t res_is_f32, const float* gamma, const float* beta, void* y,
                           int64_t ldy, void* y2, int64_t ldy2, void* y_twin, void* xhat, float* rstd, int32_t M,
                           int32_t H, float eps, float p_drop, uint64_t seed, void* stream) {
    if (!x || !gamma || !beta || !y) return ICKA_E_ARG;
    if (x_is_f32 < 0 || x_is_f32 > 2 || res_is_f32 < 0 || res_is_f32 > 2) return ICKA_E_ARG;
    if (M <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH) return ICKA_E_SHAPE;
    if (ldx % 8 || ldy % 8 || (residual && ldr % 8) || (y2 && ldy2 % 8)) return ICKA_E_ALIGN;
    if (!al16(x) || !al16(y) || (residual && !al16(residual)) || (y2 && !al16(y2)) || (xhat && !al16(xhat)) ||
        (bias && !al16(bias)) || !al16(gamma) || !al16(beta) || (y_twin && !al16(y_twin)))
        return ICKA_E_ALIGN;
    LnFwdArgs a{x, ldx, x_is_f32, bias, residual, ldr, res_is_f32, gamma, beta, (bf16_t*)y, ldy,
                (bf16_t*)y2, ldy2, y_twin, (bf16_t*)xhat, rstd, M, H, eps, make_drop(p_drop, seed), twin_f16};
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_NCH(pick_nch(H), ln_fwd_kernel, fwd_grid(M), 0, st, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_ln_fwd(const void* x, int64_t ldx, int32_t x_is_f32, const float* bias, const void* residual,
                           int64_t ldr, int32_t res_is_f32, const float* gamma, const float* beta, void* y,
                           int64_t ldy, void* y2, int64_t ldy2, float* y_f32, void* xhat, float* rstd, int32_t M,
                           int32_t H, float eps, float p_drop, uint64_t seed, void* stream) {
    return ln_fwd_impl(0, x, ldx, x_is_f32, bias, residual, ldr, res_is_f32, gamma, beta, y, ldy, y2, ldy2, y_f32, xhat, rstd,
                       M, H, eps, p_drop, seed, stream);
}
// "mixed16" form: the twin copy of the output is fp16 (forward GEMM operand + residual of the next block)
extern "C" int icka_ln_fwd_h(const void* x, int64_t ldx, int32_t x_kind, const float* bias, const void* residual,
                             int64_t ldr, int32_t res_kind, const float* gamma, const float* beta, void* y,
                             int64_t ldy, void* y2, int64_t ldy2, void* y_f16, void* xhat, float* rstd, int32_t M,
                             int32_t H, float eps, float p_drop, uint64_t seed, void* stream) {
    return ln_fwd_impl(1, x, ldx, x_kind, bias, residual, ldr, res_kind, gamma, beta, y, ldy, y2, ldy2, y_f16, xhat, rstd,
                       M, H, eps, p_drop, seed, stream);
}

extern "C" int icka_ln_bwd(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* xhat,
                           const float* rstd, const float* gamma, void* dres, int64_t lddres, void* dx, int64_t lddx,
                           float* dgamma, float* dbeta, float* dbias, float* partials, int32_t M, int32_t H,
                           float p_drop, uint64_t seed, int32_t accumulate, void* stream) {
    if (!dy || !xhat || !rstd || !gamma || !partials) return ICKA_E_ARG;
    if (dbias && !dx) return ICKA_E_ARG;   // dbias is the column sum of dx
    if (M <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH) return ICKA_E_SHAPE;
    if (lddy % 8 || (dy2 && lddy2 % 8) || (dres && lddres % 8) || (dx && lddx % 8)) return ICKA_E_ALIGN;
    if (!al16(dy) || (dy2 && !al16(dy2)) || !al16(xhat) || (dres && !al16(dres)) || (dx && !al16(dx)) || !al16(gamma))
        return ICKA_E_ALIGN;
    LnBwdArgs a{(const bf16_t*)dy, lddy, (const bf16_t*)dy2, lddy2, (const bf16_t*)xhat, rstd, gamma,
                (bf16_t*)dres, lddres, (bf16_t*)dx, lddx, partials, M, H, make_drop(p_drop, seed)};
    hipStream_t st = (hipStream_t)stream;
    int groups = (M + 31) / 32; groups = groups > COL_GROUPS ? COL_GROUPS : groups;
    const int rpg = (M + groups - 1) / groups;
    const int nx = (H + 255) / 256;
    if ((dgamma || dbeta) && !dbias) {
        const LnColsArgs c{nx, groups, rpg};
        const int grid = nx * groups + bwd_row_grid(M);
        switch (pick_nch(H)) {
            case 1: hipLaunchKernelGGL((ln_bwd_fused_kernel<1>), dim3(grid), dim3(256), 0, st, a, c); break;
            case 2: hipLaunchKernelGGL((ln_bwd_fused_kernel<2>), dim3(grid), dim3(256), 0, st, a, c); break;
            case 3: hipLaunchKernelGGL((ln_bwd_fused_kernel<3>), dim3(grid), dim3(256), 0, st, a, c); break;
            default: hipLaunchKernelGGL((ln_bwd_fused_kernel<4>), dim3(grid), dim3(256), 0, st, a, c); break;
        }
        ICKA_CHECK_LAUNCH();
        hipLaunchKernelGGL(finalize_kernel, dim3((2 * H + 63) / 64), dim3(1024), 0, st, partials, groups, H, dgamma,
                           dbeta, (float*)nullptr, (float*)nullptr, accumulate);   // slots 0, 1 only
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    DISPATCH_NCH(pick_nch(H), ln_bwd_kernel, bwd_row_grid(M), 0, st, a);
    ICKA_CHECK_LAUNCH();
    if (dgamma || dbeta || dbias) {
        hipLaunchKernelGGL(ln_cols_kernel, dim3((H + 255) / 256, groups), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)dy2, lddy2, (const bf16_t*)xhat, dbias ? (const bf16_t*)dx : nullptr, lddx,
                           partials, M, H, rpg);
        ICKA_CHECK_LAUNCH();
        hipLaunchKernelGGL(finalize_kernel, dim3((SLOTS * H + 63) / 64), dim3(1024), 0, st, partials, groups, H, dgamma,
                           dbeta, dbias, (float*)nullptr, accumulate);
        ICKA_CHECK_LAUNCH();
    }
    return 0;
}

// Row + column halves only: the column slabs partials[nslab][SLOTS][H] (slot 0 = sum dy*xhat, slot 1 = sum dy) are
// left for a later reduction (icka_gemm_grouped_ex folds it into the layer's weight-gradient launch).
extern "C" int32_t icka_ln_bwd_nslab(int32_t M) {
    int groups = (M + 31) / 32;
    return groups > COL_GROUPS ? COL_GROUPS : groups;
}
extern "C" int32_t icka_ln_slab_slots(void) { return SLOTS; }

extern "C" int icka_ln_bwd_slabs(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* xhat,
                                 const float* rstd, const float* gamma, void* dres, int64_t lddres, void* dx,
                                 int64_t lddx, float* partials, int32_t M, int32_t H, float p_drop, uint64_t seed,
                                 void* stream) {
    if (!dy || !xhat || !rstd || !gamma || !partials) return ICKA_E_ARG;
    if (M <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH) return ICKA_E_SHAPE;
    if (lddy % 8 || (dy2 && lddy2 % 8) || (dres && lddres % 8) || (dx && lddx % 8)) return ICKA_E_ALIGN;
    if (!al16(dy) || (dy2 && !al16(dy2)) || !al16(xhat) || (dres && !al16(dres)) || (dx && !al16(dx)) || !al16(gamma))
        return ICKA_E_ALIGN;
    LnBwdArgs a{(const bf16_t*)dy, lddy, (const bf16_t*)dy2, lddy2, (const bf16_t*)xhat, rstd, gamma,
                (bf16_t*)dres, lddres, (bf16_t*)dx, lddx, partials, M, H, make_drop(p_drop, seed)};
    const int groups = icka_ln_bwd_nslab(M);
    const int nx = (H + 255) / 256;
    const LnColsArgs c{nx, groups, (M + groups - 1) / groups};
    const int grid = nx * groups + bwd_row_grid(M);
    hipStream_t st = (hipStream_t)stream;
    switch (pick_nch(H)) {
        case 1: hipLaunchKernelGGL((ln_bwd_fused_kernel<1>), dim3(grid), dim3(256), 0, st, a, c); break;
        case 2: hipLaunchKernelGGL((ln_bwd_fused_kernel<2>), dim3(grid), dim3(256), 0, st, a, c); break;
        case 3: hipLaunchKernelGGL((ln_bwd_fused_kernel<3>), dim3(grid), dim3(256), 0, st, a, c); break;
        default: hipLaunchKernelGGL((ln_bwd_fused_kernel<4>), dim3(grid), dim3(256), 0, st, a, c); break;
    }
    ICKA_CHECK_LAUNCH();
    return 0;
}

static int embed_fwd_impl(int32_t twin_f16, const int64_t* ids, const int64_t* token_type, const float* word, const float* pos,
                              const float* type, const float* gamma, const float* beta, void* y, void* y_twin,
                              void* xhat, float* rstd, int32_t B, int32_t S, int32_t H, int32_t vocab,
                              int32_t n_type, float eps, float p_drop, uint64_t seed, void* stream) {
    if (!ids || !word || !pos || !type || !gamma || !beta || !y) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH || vocab <= 0 || n_type <= 0)
        return ICKA_E_SHAPE;
    if (!al16(word) || !al16(pos) || !al16(type) || !al16(gamma) || !al16(beta) || !al16(y) || (xhat && !al16(xhat)) ||
        (y_twin && !al16(y_twin)))
        return ICKA_E_ALIGN;
    EmbFwdArgs a{ids, token_type, word, pos, type, gamma, beta, (bf16_t*)y, y_twin, (bf16_t*)xhat, rstd,
                 B * S, S, H, vocab, n_type, eps, make_drop(p_drop, seed), nullptr, nullptr, S, 0, 0, twin_f16};
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_NCH(pick_nch(H), embed_fwd_kernel, row_grid(B * S), 0, st, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_embed_fwd(const int64_t* ids, const int64_t* token_type, const float* word, const float* pos,
                              const float* type, const float* gamma, const float* beta, void* y, float* y_f32,
                              void* xhat, float* rstd, int32_t B, int32_t S, int32_t H, int32_t vocab,
                              int32_t n_type, float eps, float p_drop, uint64_t seed, void* stream) {
    return embed_fwd_impl(0, ids, token_type, word, pos, type, gamma, beta, y, y_f32, xhat, rstd, B, S, H, vocab, n_type, eps,
                          p_drop, seed, stream);
}
// "mixed16" form: fp16 twin of the output
extern "C" int icka_embed_fwd_h(const int64_t* ids, const int64_t* token_type, const float* word, const float* pos,
                                const float* type, const float* gamma, const float* beta, void* y, void* y_f16,
                                void* xhat, float* rstd, int32_t B, int32_t S, int32_t H, int32_t vocab,
                                int32_t n_type, float eps, float p_drop, uint64_t seed, void* stream) {
    return embed_fwd_impl(1, ids, token_type, word, pos, type, gamma, beta, y, y_f16, xhat, rstd, B, S, H, vocab, n_type, eps,
                          p_drop, seed, stream);
}

// Embeddings of a prompt-spliced sequence (the prompt-accepting encoder stage of the current reference model,
// Cross_Modal_Interaction_Module.py:1010-1012): out[b, t] = LN(x + pos[t + pos_offset] + type[0]) with
// x = word[ids[b, src[t]]] for src[t] >= 0, prompt[b, -1-src[t]] otherwise; S = len(src) output positions.
extern "C" int icka_embed_prompt_fwd(const int64_t* ids, const int32_t* src, const void* prompt, const float* word,
                                     const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                                     float* y_f32, void* xhat, float* rstd, int32_t B, int32_t S_in, int32_t S,
                                     int32_t P, int32_t H, int32_t vocab, int32_t pos_offset, float eps, float p_drop,
                                     uint64_t seed, void* stream) {
    if (!ids || !src || !prompt || !word || !pos || !type || !gamma || !beta || !y) return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || S_in <= 0 || P <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH || vocab <= 0 || pos_offset < 0)
        return ICKA_E_SHAPE;
    if (!al16(word) || !al16(pos) || !al16(type) || !al16(gamma) || !al16(beta) || !al16(y) || (xhat && !al16(xhat)) ||
        !al16(prompt))
        return ICKA_E_ALIGN;
    EmbFwdArgs a{ids, nullptr, word, pos, type, gamma, beta, (bf16_t*)y, y_f32, (bf16_t*)xhat, rstd,
                 B * S, S, H, vocab, 1, eps, make_drop(p_drop, seed), src, (const bf16_t*)prompt, S_in, P, pos_offset, 0};
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_NCH(pick_nch(H), embed_fwd_kernel, row_grid(B * S), 0, st, a);
    ICKA_CHECK_LAUNCH();
    return 0;
}

static int embed_bwd_impl(const void* dy, const int64_t* ids, const int64_t* token_type, const void* xhat,
                          const float* rstd, const float* gamma, float* dword, float* dtok, float* dpos, float* dtype,
                          float* dgamma, float* dbeta, float* partials, int32_t B, int32_t S, int32_t H,
                          int32_t vocab, int32_t n_type, int32_t padding_idx, float p_drop, uint64_t seed,
                          int32_t accumulate, void* stream) {
    if (!dy || !ids || !xhat || !rstd || !gamma || (!dword && !dtok) || !dpos || !dtype || !dgamma || !dbeta || !partials)
        return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH || vocab <= 0 || n_type <= 0)
        return ICKA_E_SHAPE;
    if (!al16(dy) || !al16(xhat) || !al16(gamma)) return ICKA_E_ALIGN;
    EmbBwdArgs a{(const bf16_t*)dy, ids, token_type, (const bf16_t*)xhat, rstd, gamma, dword, dpos, dtype, partials,
                 B * S, S, H, vocab, n_type, padding_idx, make_drop(p_drop, seed), nullptr, nullptr, S, 0, 0, dtok};
    hipStream_t st = (hipStream_t)stream;
    int grid = bwd_grid(B * S);
    if (S <= BWD_BLOCKS) {   // position-major: one block per position, S slabs
        grid = S;
        switch (pick_nch(H)) {
            case 1: hipLaunchKernelGGL((embed_bwd_pos_kernel<1>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
            case 2: hipLaunchKernelGGL((embed_bwd_pos_kernel<2>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
            case 3: hipLaunchKernelGGL((embed_bwd_pos_kernel<3>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
            default: hipLaunchKernelGGL((embed_bwd_pos_kernel<4>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
        }
    } else {
        DISPATCH_NCH(pick_nch(H), embed_bwd_kernel, grid, 8 * H * sizeof(float), st, a);
    }
    ICKA_CHECK_LAUNCH();
    float* t0 = n_type <= 2 ? dtype : nullptr;
    float* t1 = n_type == 2 ? dtype + H : nullptr;
    hipLaunchKernelGGL(finalize_kernel, dim3((SLOTS * H + 63) / 64), dim3(1024), 0, st, partials, grid, H, dgamma,
                       dbeta, t0, t1, accumulate);
    ICKA_CHECK_LAUNCH();
    return 0;
}

extern "C" int icka_embed_bwd(const void* dy, const int64_t* ids, const int64_t* token_type, const void* xhat,
                              const float* rstd, const float* gamma, float* dword, float* dpos, float* dtype,
                              float* dgamma, float* dbeta, float* partials, int32_t B, int32_t S, int32_t H,
                              int32_t vocab, int32_t n_type, int32_t padding_idx, float p_drop, uint64_t seed,
                              int32_t accumulate, void* stream) {
    if (!dword) return ICKA_E_ARG;
    return embed_bwd_impl(dy, ids, token_type, xhat, rstd, gamma, dword, nullptr, dpos, dtype, dgamma, dbeta, partials, B, S, H, vocab,
                          n_type, padding_idx, p_drop, seed, accumulate, stream);
}
// Row-sparse form for the data-parallel exchange (dp.GradReducer(sparse_embeddings=True)): everything as icka_embed_bwd, except
// that the word-table gradient is left as per-token rows dtok f32 [B*S, H] (zero rows for the padding id) for an all-gather
// + icka_embed_scatter_rows instead of being scattered into the [vocab, H] table here.
extern "C" int icka_embed_bwd_rows(const void* dy, const int64_t* ids, const int64_t* token_type, const void* xhat,
                                   const float* rstd, const float* gamma, float* dtok, float* dpos, float* dtype,
                                   float* dgamma, float* dbeta, float* partials, int32_t B, int32_t S, int32_t H,
                                   int32_t vocab, int32_t n_type, int32_t padding_idx, float p_drop, uint64_t seed,
                                   int32_t accumulate, void* stream) {
    if (!dtok) return ICKA_E_ARG;
    return embed_bwd_impl(dy, ids, token_type, xhat, rstd, gamma, nullptr, dtok, dpos, dtype, dgamma, dbeta, partials, B, S, H, vocab,
                          n_type, padding_idx, p_drop, seed, accumulate, stream);
}

namespace {
// dword[ids[t]] += scale * rows[t] for T gathered token rows (f32 or bf16 rows); a wave per row, lane-strided f32 atomics
// (64 consecutive floats per wave-instruction), rows of the padding id (and ids outside the table) skipped.
__global__ __launch_bounds__(256) void embed_scatter_rows_kernel(const void* __restrict__ rows, int rows_bf16,
                                                                 const int64_t* __restrict__ ids, float* __restrict__ dword,
                                                                 int64_t T, int H, int vocab, int padding_idx, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    for (int64_t t = w; t < T; t += nw) {
        const int64_t id = ids[t];
        if (id == padding_idx || id < 0 || id >= vocab) continue;
        float* dst = dword + id * H;
        if (rows_bf16) {
            const bf16_t* r = reinterpret_cast<const bf16_t*>(rows) + t * H;
            for (int j = lane; j < H; j += 64) atomicAdd(dst + j, scale * bf2f(r[j]));
        } else {
            const float* r = reinterpret_cast<const float*>(rows) + t * H;
            for (int j = lane; j < H; j += 64) atomicAdd(dst + j, scale * r[j]);
        }
    }
}
}  // namespace

extern "C" int icka_embed_scatter_rows(const void* rows, int32_t rows_are_bf16, const int64_t* ids, float* dword, int64_t T,
                                       int32_t H, int32_t vocab, int32_t padding_idx, float scale, void* stream) {
    if (!rows || !ids || !dword) return ICKA_E_ARG;
    if (T < 0 || H <= 0 || vocab <= 0) return ICKA_E_SHAPE;
    if (T == 0) return 0;
    int64_t blocks = (T + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(embed_scatter_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, rows_are_bf16, ids,
                       dword, T, H, vocab, padding_idx, scale);
    ICKA_CHECK_LAUNCH();
    return 0;
}

// Backward of icka_embed_prompt_fwd: word / position / type-row-0 table gradients, dgamma, dbeta, and the gradient of
// the prompt block (bf16 [B,P,H], every row written exactly once).  partials: S * 4 * H floats.
extern "C" int icka_embed_prompt_bwd(const void* dy, const int64_t* ids, const int32_t* src, const void* xhat,
                                     const float* rstd, const float* gamma, float* dword, float* dpos, float* dtype,
                                     float* dgamma, float* dbeta, void* dprompt, float* partials, int32_t B,
                                     int32_t S_in, int32_t S, int32_t P, int32_t H, int32_t vocab, int32_t pos_offset,
                                     int32_t padding_idx, float p_drop, uint64_t seed, int32_t accumulate, void* stream) {
    if (!dy || !ids || !src || !xhat || !rstd || !gamma || !dword || !dpos || !dtype || !dgamma || !dbeta || !dprompt ||
        !partials)
        return ICKA_E_ARG;
    if (B <= 0 || S <= 0 || S > BWD_BLOCKS || S_in <= 0 || P <= 0 || H <= 0 || H % 8 != 0 || H > 8 * 64 * MAX_CH ||
        vocab <= 0 || pos_offset < 0)
        return ICKA_E_SHAPE;
    if (!al16(dy) || !al16(xhat) || !al16(gamma)) return ICKA_E_ALIGN;
    EmbBwdArgs a{(const bf16_t*)dy, ids, nullptr, (const bf16_t*)xhat, rstd, gamma, dword, dpos, dtype, partials,
                 B * S, S, H, vocab, 1, padding_idx, make_drop(p_drop, seed), src, (bf16_t*)dprompt, S_in, P, pos_offset};
    hipStream_t st = (hipStream_t)stream;
    switch (pick_nch(H)) {
        case 1: hipLaunchKernelGGL((embed_bwd_pos_kernel<1>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
        case 2: hipLaunchKernelGGL((embed_bwd_pos_kernel<2>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
        case 3: hipLaunchKernelGGL((embed_bwd_pos_kernel<3>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
        default: hipLaunchKernelGGL((embed_bwd_pos_kernel<4>), dim3(S), dim3(512), 8 * H * sizeof(float), st, a); break;
    }
    ICKA_CHECK_LAUNCH();
    hipLaunchKernelGGL(finalize_kernel, dim3((SLOTS * H + 63) / 64), dim3(1024), 0, st, partials, S, H, dgamma, dbeta,
                       dtype, (float*)nullptr, accumulate);   // slot 2 = token-type row 0 (every row is type 0)
    ICKA_CHECK_LAUNCH();
    return 0;
}
