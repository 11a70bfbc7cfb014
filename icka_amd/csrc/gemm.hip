// MFMA GEMM for the ICKA hot path (gfx950), v_mfma_f32_16x16x32_bf16, fp32 accumulation.  Three operand layouts:
//   NT: C = A[M,K] . B[N,K]^T   (both operands k-contiguous: fragments by ds_read_b128)
//   NN: C = A[M,K] . B[K,N]     (B k-major in memory: fragments by ds_read_b64_tr_b16)
//   TN: C = A[K,M]^T . B[K,N]   (both k-major: weight gradients, K = tokens)
// The MFMA is issued as D^T: a-operand = B-tile rows (n), b-operand = A-tile rows (m), so each lane ends up with
// FOUR CONSECUTIVE n of one output row m -> 8-byte (bf16) / 16-byte (f32) row-major stores and vector epilogues.
// Kernels in this file:
//   gemm_kernel            general path: any M, N, K / ragged / unaligned operands, register-staged, split-K
//   gemm_dma_kernel        aligned path, 256 threads, LDS-DMA staging (predecessor of the warp-specialised kernels)
//   gemm_ws_kernel         aligned path, 512 threads: 4 loader waves (LDS-DMA) + 4 compute waves, ring of 3 k-tiles,
//                          128x128 or 128x96 output tiles, LDS-staged or direct (f32) epilogue
//   gemm_ws2_kernel        same with a ring of 2 and two co-resident blocks per CU (short-K, many-tile grids)
//   gemm_ws_group_kernel   up to 4 problems of one layout in one launch (128x128 tiles)
//   gemm_w3_kernel         12 waves (8 compute + 4 loader), 256x192 tiles (wide short-K outputs: qkv, ffn-up, d-ffn-down) or
//                          256x128 tiles (N = 1024 / 768 at M = 8192: one round of 192..256 tiles instead of two)
//   gemm_big_group_kernel  256x128 tiles for the grouped weight gradients + column-sum / slab-reduction blocks
// Variants of the NT kernels: F16 (IEEE fp16 operands on v_mfma_f32_16x16x32_f16, fp16 / bf16-copy outputs: the "mixed16"
// forward GEMMs) and CONV (gemm_ws_kernel only: the A tile is gathered from an NHWC activation = implicit 3x3 convolution).
// LDS images are XOR-swizzled per layout (off_kc / off_km), the swizzle applied to the DMA source address.
#include "common.h"
#include "ln_row.h"
#include "attn_core.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = 128 * 64 * 2;  // one operand tile, either layout

struct GemmArgs {
    int M, N, K, K1;
    const bf16_t* A;  int64_t lda;
    const bf16_t* B;  int64_t ldb;
    const bf16_t* A2; int64_t lda2;
    const bf16_t* B2; int64_t ldb2;
    void* C;  int64_t ldc;
    bf16_t* C2; int64_t ldc2;
    const bf16_t* aux; int64_t ldaux;
    const float* bias; const float* bias2;
    float alpha, beta;
    int epi, c_f32;
    int a_vec, b_vec;  // operand rows may be read with 16-byte loads (ld % 8 == 0, base 16-B aligned)
    int abl;           // diagnostic ablation of the fast path: 1 = no MFMA/LDS reads, 2 = no DMA staging
    float* colsum; int colsum_acc;  // TN fast path: colsum[m] (+)= sum_k A[k,m] (bias gradient fused into dW = dY^T.X)
    int n96ok;                  // fast path + N % 96 == 0: the 128x96 tile is an option
    int n64ok;                  // everything aligned except N % 128: N % 64 == 0 -> the 128x64 tile (NT only)
    int direct;                 // plain outputs skip the LDS-staged epilogue (Tune::direct)
    unsigned long long* stamp;  // diagnostic: [block][8] cycle sums (ICKA_GEMM_STAMP builds)
    int ksplit;        // general path: blockIdx.y splits the k-tiles; partial sums are atomically added to f32 C
    int f16;           // operands are IEEE fp16 (v_mfma_f32_16x16x32_f16; NT only): the "mixed16" forward GEMMs
    int c_f16;         // main output C is fp16 (c_f32 == 0)
    bf16_t* C3; int64_t ldc3;   // optional bf16 copy of the main output: of an fp16 activation (the weight-gradient operand of
                                // the "mixed16" mode), or of an f32 weight gradient (the data-parallel wire copy: the value
                                // AFTER beta-accumulation, written by the same epilogue -- dp.GradReducer)
    int aux_f16;       // the epilogue operand aux is fp16 (read through load8_aux / load4_aux)
    int w3_sn, w3_pnlog;   // 12-wave kernel: column tiles per XCD patch and log2(pn) of the pm x pn XCD cut (gemm_w3_grid)
    int c3_only;       // f32 C + C3 + beta == 0: write ONLY the bf16 wire copy C3 (the f32 value is produced later, from the
                       // reduced wire buffer, by icka_dp_cast_back_scaled): the epilogue stores 2 bytes per element, not 4 + 2
    int plain;         // the MAIN 16-bit output is stored with ordinary stores instead of streaming ones (st_main): per launch,
                       // from the diagnostic site mask ICKA_GEMM_PLAIN_MASK (site_bit)
    // implicit 3x3 / pad 1 convolution (icka_conv3x3_gemm): A is an NHWC activation [B, cvH, cvW, cvC], the A "row" m is
    // output pixel m and the reduction index is k = tap * cvC + c -- the loader waves compute the patch addresses, no
    // patch matrix exists.  cvZero: at least 128 B of zeros for the taps that fall off the image / rows past cvRows.
    int cvH, cvW, cvC, cvS, cvHo, cvWo, cvRows;
    const bf16_t* cvZero;
};

// k-contiguous tile image [128 rows][64 k]: 128-B rows, 16-B chunk index XORed with (row>>1)&7 so that the 16 rows
// of a fragment read land on 16 distinct 16-B slots of the 256-B bank row.
__device__ __forceinline__ uint32_t off_kc(int row, int ch) { return row * 128 + (((ch ^ (row >> 1)) & 7) << 4); }
// k-major tile image [64 k][128 rows]: 256-B rows; chunk XOR per cdna guide T10 image (b), serves the transposed read.
__device__ __forceinline__ uint32_t off_km(int kr, int ch) {
    return kr * 256 + ((ch ^ (((kr & 3) << 2) | ((kr >> 2) & 3))) << 4);
}

__device__ __forceinline__ u32x4 load_partial(const bf16_t* p, int nvalid) {
    const unsigned short* ps = reinterpret_cast<const unsigned short*>(p);
    u32x4 v = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t x = e < nvalid ? (uint32_t)ps[e] : 0u;
        v[e >> 1] |= x << (16 * (e & 1));
    }
    return v;
}

// global -> registers: 4 x 16-byte chunks per thread cover the 128x64 (or 64x128) tile.
template <bool KM, bool ALIGNED>
__device__ __forceinline__ void g2r(u32x4 (&r)[4], const bf16_t* __restrict__ P, int64_t ld, int row0, int nrows,
                                    int k0, int K, int tid, int vec) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        const bf16_t* p;
        bool ok;
        int rem;
        if (!KM) {
            const int rr = q >> 3, c = q & 7;
            const int grow = row0 + rr, gk = k0 + 8 * c;
            p = P + (int64_t)grow * ld + gk;
            ok = grow < nrows;
            rem = K - gk;
        } else {
            const int kr = q >> 4, c = q & 15;
            const int gk = k0 + kr, gcol = row0 + 8 * c;
            p = P + (int64_t)gk * ld + gcol;
            ok = gk < K;
            rem = nrows - gcol;
        }
        if (ALIGNED) {
            r[i] = *reinterpret_cast<const u32x4*>(p);
        } else {
            if (ok && rem >= 8 && vec) r[i] = *reinterpret_cast<const u32x4*>(p);
            else if (ok && rem > 0) r[i] = load_partial(p, rem < 8 ? rem : 8);
            else r[i] = u32x4{0u, 0u, 0u, 0u};
        }
    }
}

template <bool KM>
__device__ __forceinline__ void r2s(const u32x4 (&r)[4], char* tile, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i;
        const uint32_t off = KM ? off_km(q >> 4, q & 15) : off_kc(q >> 3, q & 7);
        *reinterpret_cast<u32x4*>(tile + off) = r[i];
    }
}

// MFMA operand fragment for the 16 tile-rows starting at rbase, k-step ks (32 k each) of the 64-deep tile:
// lane l gets element e = operand[row rbase + (l&15)][k = 32*ks + 8*(l>>4) + e].
template <bool KM>
__device__ __forceinline__ bf16x8 read_frag(const char* tile, int rbase, int ks, int lane) {
    if (!KM) {
        return lds_read_b128(tile, off_kc(rbase + (lane & 15), ks * 4 + (lane >> 4)));
    } else {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int c = (rbase >> 3) + (p >> 1);
        const int kr = ks * 32 + 8 * g + q;
        const bf16x4 lo = lds_read_tr(tile, off_km(kr, c) + 8 * (p & 1));
        const bf16x4 hi = lds_read_tr(tile, off_km(kr + 4, c) + 8 * (p & 1));
        return join8(lo, hi);
    }
}

__device__ __forceinline__ void load4(const bf16_t* base, int64_t ld, int m, int n, int nvalid, float (&o)[4]) {
    const bf16_t* p = base + (int64_t)m * ld + n;
    if (nvalid == 4 && ((reinterpret_cast<uintptr_t>(p) & 7) == 0)) {
        const bf16x4 v = as_bf16x4(*reinterpret_cast<const u32x2*>(p));
        o[0] = bf2f(v[0]); o[1] = bf2f(v[1]); o[2] = bf2f(v[2]); o[3] = bf2f(v[3]);
    } else {
        for (int r = 0; r < 4; ++r) o[r] = r < nvalid ? bf2f(p[r]) : 0.f;
    }
}
__device__ __forceinline__ void load4_aux(const bf16_t* base, int64_t ld, int m, int n, int nvalid, int f16, float (&o)[4]) {
    if (!f16) { load4(base, ld, m, n, nvalid, o); return; }
    const _Float16* p = reinterpret_cast<const _Float16*>(base) + (int64_t)m * ld + n;
    for (int r = 0; r < 4; ++r) o[r] = r < nvalid ? (float)p[r] : 0.f;
}
__device__ __forceinline__ void store4_bf16(bf16_t* base, int64_t ld, int m, int n, int nvalid, const float (&v)[4]) {
    bf16_t* p = base + (int64_t)m * ld + n;
    if (nvalid == 4 && ((reinterpret_cast<uintptr_t>(p) & 7) == 0)) {
        *reinterpret_cast<u32x2*>(p) = pack4(v[0], v[1], v[2], v[3]);
    } else {
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) p[r] = f2bf(v[r]);
    }
}

__device__ __forceinline__ void epilogue4(const GemmArgs& g, int m, int n, f32x4 acc) {
    const int nvalid = (g.N - n) < 4 ? (g.N - n) : 4;
    float v[4] = {acc[0] * g.alpha, acc[1] * g.alpha, acc[2] * g.alpha, acc[3] * g.alpha};
    if (g.bias && blockIdx.y == 0) {
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) v[r] += g.bias[n + r];
    }
    if (g.bias2 && blockIdx.y == 0) {
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) v[r] += g.bias2[n + r];
    }
    float a[4];
    switch (g.epi) {
        case ICKA_EPI_GELU:
            store4_bf16(g.C2, g.ldc2, m, n, nvalid, v);
            for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            break;
        case ICKA_EPI_DGELU:
            load4_aux(g.aux, g.ldaux, m, n, nvalid, g.aux_f16, a);
            for (int r = 0; r < 4; ++r) v[r] *= dgelu_f(a[r]);
            break;
        case ICKA_EPI_ADD:
            load4_aux(g.aux, g.ldaux, m, n, nvalid, g.aux_f16, a);
            for (int r = 0; r < 4; ++r) v[r] += a[r];
            break;
        case ICKA_EPI_GATE:
            for (int r = 0; r < 4; ++r) v[r] = sigmoid_f(v[r]);
            if (g.C2) store4_bf16(g.C2, g.ldc2, m, n, nvalid, v);
            load4_aux(g.aux, g.ldaux, m, n, nvalid, g.aux_f16, a);
            for (int r = 0; r < 4; ++r) v[r] *= a[r];
            break;
        case ICKA_EPI_TANH:
            for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
            break;
        case ICKA_EPI_ADD_RELU:
            load4_aux(g.aux, g.ldaux, m, n, nvalid, g.aux_f16, a);
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r] + a[r], 0.f);
            break;
        case ICKA_EPI_RELU:
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            break;
        default: break;
    }
    if (g.ksplit > 1) {   // partial sum of one k-range (host guarantees f32 C, plain epilogue, C pre-scaled by beta)
        float* p = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n;
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) atomicAdd(p + r, v[r]);
        return;
    }
    if (g.c_f32 && g.c3_only) {
        store4_bf16(g.C3, g.ldc3, m, n, nvalid, v);
    } else if (g.c_f32) {
        float* p = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n;
        if (nvalid == 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            if (g.beta != 0.f) o += g.beta * *reinterpret_cast<const f32x4*>(p);
            *reinterpret_cast<f32x4*>(p) = o;
            for (int r = 0; r < 4; ++r) v[r] = o[r];
        } else {
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) { v[r] += g.beta != 0.f ? g.beta * p[r] : 0.f; p[r] = v[r]; }
        }
        if (g.C3) store4_bf16(g.C3, g.ldc3, m, n, nvalid, v);
    } else if (g.c_f16) {
        _Float16* p = reinterpret_cast<_Float16*>(g.C) + (int64_t)m * g.ldc + n;
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) p[r] = (_Float16)fminf(fmaxf(v[r], -65504.f), 65504.f);
        if (g.C3) store4_bf16(g.C3, g.ldc3, m, n, nvalid, v);
    } else {
        bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
        if (g.beta != 0.f) {
            load4(C, g.ldc, m, n, nvalid, a);
            for (int r = 0; r < 4; ++r) v[r] += g.beta * a[r];
        }
        store4_bf16(C, g.ldc, m, n, nvalid, v);
    }
}

// Issue the global loads of k-tile kt for both operands (second reduction segment after K1, if any).
template <bool A_KM, bool B_KM, bool ALIGNED>
__device__ __forceinline__ void fetch_tiles(const GemmArgs& g, int kt, int m0, int n0, int tid, u32x4 (&ra)[4],
                                            u32x4 (&rb)[4]) {
    int k0 = kt * BK;
    const bf16_t* Ap = g.A; int64_t la = g.lda;
    const bf16_t* Bp = g.B; int64_t lb = g.ldb;
    int klim = g.K1 > 0 ? g.K1 : g.K;
    if (g.K1 > 0 && k0 >= g.K1) {
        Ap = g.A2; la = g.lda2; Bp = g.B2; lb = g.ldb2;
        k0 -= g.K1; klim = g.K - g.K1;
    }
    g2r<A_KM, ALIGNED>(ra, Ap, la, m0, g.M, k0, klim, tid, g.a_vec);
    g2r<B_KM, ALIGNED>(rb, Bp, lb, n0, g.N, k0, klim, tid, g.b_vec);
}

template <bool A_KM, bool B_KM, bool ALIGNED, bool F16 = false>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmArgs gp) {
    const GemmArgs g = gp;  // local copy: lets SROA scalarise the descriptor instead of spilling the kernarg struct
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [2 buffers][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;

    // XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous
    // run of tiles (same A row panel, neighbouring B panels) -> its private L2 sees the panel re-use.
    const int nbn = (g.N + BN - 1) / BN;
    const int nb = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, qn = nb >> 3, rn = nb & 7;
    const int sw = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + (bid >> 3);
    const int m0 = (sw / nbn) * BM, n0 = (sw % nbn) * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk_all = (g.K + BK - 1) / BK;
    // split-K (skinny outputs with a long reduction): blockIdx.y owns a contiguous range of k-tiles
    const int per = (nk_all + g.ksplit - 1) / g.ksplit;
    const int kt0 = blockIdx.y * per;
    const int nk = (kt0 + per < nk_all ? kt0 + per : nk_all) - kt0;
    if (nk <= 0) return;
    u32x4 ra[4], rb[4];

    fetch_tiles<A_KM, B_KM, ALIGNED>(g, kt0, m0, n0, tid, ra, rb);
    r2s<A_KM>(ra, smem, tid);
    r2s<B_KM>(rb, smem + TILE_BYTES, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const char* sA = smem + (kt & 1) * 2 * TILE_BYTES;
        const char* sB = sA + TILE_BYTES;
        if (kt + 1 < nk) fetch_tiles<A_KM, B_KM, ALIGNED>(g, kt0 + kt + 1, m0, n0, tid, ra, rb);  // under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = read_frag<A_KM>(sA, wr + 16 * t, ks, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t] = read_frag<B_KM>(sB, wc + 16 * t, ks, lane);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16t<F16>(fb[ni], fa[mi], acc[mi][ni]);
        }
        if (kt + 1 < nk) {
            char* dA = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
            r2s<A_KM>(ra, dA, tid);
            r2s<B_KM>(rb, dA + TILE_BYTES, tid);
        }
        __syncthreads();
    }

    // D^T layout: lane holds C[m = .. + (lane&15)][n = .. + 4*(lane>>4) + r]
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int m = m0 + wr + 16 * mi + (lane & 15);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = n0 + wc + 16 * ni + 4 * (lane >> 4);
            if (m < g.M && n < g.N) epilogue4(g, m, n, acc[mi][ni]);
        }
    }
}


// =====================================================================================================================
// Fast path (M % 128 == 0, N % 128 == 0, K % 64 == 0, 16-B aligned operands): same tile / MFMA / LDS images as above,
// but (1) tiles are staged by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction straight into LDS, no
// staging VGPRs, no ds_write pass) -- the swizzle is applied to the per-lane SOURCE address because the DMA
// destination is lane-linear -- with the next k-tile's DMA issued before the current tile's MFMAs, one barrier per
// k-tile; (2) the epilogue goes through LDS as an f32 [128][128] tile (XOR-swizzled 16-B chunks) so that every
// thread finishes 8 consecutive columns of a row: bias / activation / fan-in operands and the C stores are all
// 16-byte, row-contiguous accesses (whole 256-B rows per 16 lanes) instead of 8-byte stores scattered over 16 rows.
// Per-lane SOURCE pointers of the 4 one-KiB pieces a wave stages per operand tile (k-tile 0).  They advance by a
// wave-uniform stride per k-tile, so the k-loop carries no address arithmetic beyond 8 pointer bumps.
// NCOL = operand rows (k-contiguous) / columns (k-major) the tile really has: 128, or 96 for the narrow-N tile.  The
// LDS image keeps the 128-wide geometry; with 96 the k-contiguous image simply has no pieces 12..15 and the k-major
// image leaves chunks 12..15 of every row unused (their lanes re-fetch a valid chunk of the same row instead).
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool KM, int NCOL = 128>
__device__ __forceinline__ void dma_init(const bf16_t* (&ptr)[4], const bf16_t* __restrict__ P, int64_t ld, int row0,
                                         int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave + 4 * j;  // 1-KiB piece: 8 tile rows (k-contiguous image) or 4 k-rows (k-major image)
        if (!KM) {
            const int row = (8 * p + (lane >> 3)) % NCOL;   // pieces past NCOL/8 are never issued
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            ptr[j] = P + (int64_t)(row0 + row) * ld + 8 * lc;
        } else {
            const int kr = 4 * p + (lane >> 4);
            int lc = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
            if (lc >= NCOL / 8) lc &= 7;
            ptr[j] = P + (int64_t)kr * ld + row0 + 8 * lc;
        }
    }
}
// Issue the 4 LDS-DMA loads of one operand tile (pieces wave, wave+4, wave+8, wave+12 -> 4 KiB apart) from INLINE
// ASM: hipcc treats a compiler-visible LDS-DMA as a pending LDS store and puts `s_waitcnt vmcnt(0)` in front of the
// next ds_read_b64_tr_b16, which serialises the whole pipeline; hidden in asm, the ring is ordered only by our own
// counted vmcnt + s_barrier (cdna guide section 5.7).  M0 (LDS destination base) is saved/restored inside.
#ifndef ICKA_DMA_POL
#define ICKA_DMA_POL ""   // cache-policy bits of the operand loads (diagnostic builds: " nt", " sc1", " sc0 sc1")
#endif
template <int NJ = 4>
__device__ __forceinline__ void dma_issue(const bf16_t* (&ptr)[4], int64_t stride, uint32_t lds_base) {
    uint32_t keep;
    if constexpr (NJ == 2) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off" ICKA_DMA_POL "\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %2, off" ICKA_DMA_POL "\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(ptr[0]), "v"(ptr[1]), "s"(lds_base)
            : "memory", "scc");
#pragma unroll
        for (int j = 0; j < 2; ++j) ptr[j] += stride;
        return;
    }
    if constexpr (NJ == 3) {
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %4\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off" ICKA_DMA_POL "\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %2, off" ICKA_DMA_POL "\n\t"
            "s_add_u32 m0, m0, 0x1000\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %3, off" ICKA_DMA_POL "\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "s"(lds_base)
            : "memory", "scc");
#pragma unroll
        for (int j = 0; j < 3; ++j) ptr[j] += stride;
        return;
    }
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off" ICKA_DMA_POL "\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %2, off" ICKA_DMA_POL "\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, off" ICKA_DMA_POL "\n\t"
        "s_add_u32 m0, m0, 0x1000\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, off" ICKA_DMA_POL "\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(ptr[0]), "v"(ptr[1]), "v"(ptr[2]), "v"(ptr[3]), "s"(lds_base)
        : "memory", "scc");
#pragma unroll
    for (int j = 0; j < 4; ++j) ptr[j] += stride;
}

// f32 C tile in LDS: [128 rows][32 chunks of 4 floats], chunk index XORed with row&31
__device__ __forceinline__ uint32_t off_c(int row, int ch) { return row * 512 + ((ch ^ (row & 31)) << 4); }
// same for a 192-column f32 tile (48 chunks of 16 B per row; the XOR stays inside each group of 16 chunks)
__device__ __forceinline__ uint32_t off_cw(int row, int ch) { return row * 768 + ((ch ^ (row & 15)) << 4); }

// Output stores of the fast-path epilogues are NON-TEMPORAL: an FFN launch writes 25-50 MB, which as ordinary stores
// pushed the operand panels out of the XCD's 4 MiB L2 (rocprofv3: L2 hit rate 0.57-0.79, 3-6x the algorithmic bytes
// fetched over the fabric).  Measured on the c2 step: GEMM class 559 -> 593 TFLOP/s.
template <typename T>
__device__ __forceinline__ void st_out(T* p, const T v) {
#ifdef ICKA_ST_TEMPORAL
    *p = v;   // (diagnostic build: plain stores)
#else
    __builtin_nontemporal_store(v, p);
#endif
}
// epilogue operands that are read exactly once (GELU' input, residual / fan-in addend, accumulate-into output)
template <typename T>
__device__ __forceinline__ T ld_once(const T* p) {
#ifdef ICKA_NT_AUX
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void load8_bf16(const bf16_t* p, float (&o)[8]) {
    const bf16x8 v = as_bf16x8(ld_once(reinterpret_cast<const u32x4*>(p)));
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = bf2f(v[e]);
}
// epilogue operand in either 16-bit type (same element size: the pointer arithmetic is shared)
__device__ __forceinline__ void load8_aux(const bf16_t* p, int f16, float (&o)[8]) {
    if (f16) {
        const f16x8 v = __builtin_bit_cast(f16x8, ld_once(reinterpret_cast<const u32x4*>(p)));
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    } else {
        load8_bf16(p, o);
    }
}
// the same in two steps: the 16 raw bytes (issued early), converted where they are used
__device__ __forceinline__ void cvt8_aux(const u32x4 raw, int f16, float (&o)[8]) {
    if (f16) {
        const f16x8 v = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
    } else {
        const bf16x8 v = as_bf16x8(raw);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f(v[e]);
    }
}
__device__ __forceinline__ void store8_bf16(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    st_out(reinterpret_cast<u32x4*>(p), as_u32x4(o));
}
// MAIN activation output with a per-launch store policy (GemmArgs.plain: the ICKA_GEMM_PLAIN_MASK diagnostic): plain = an ordinary
// store, the line stays in the L2 / Infinity Cache for the consumer kernel; else the streaming store of st_out
template <typename T>
__device__ __forceinline__ void st_main(T* p, const T v, int plain) {
    if (plain) *p = v;
    else st_out(p, v);
}
__device__ __forceinline__ void store8_bf16_main(bf16_t* p, const float (&v)[8], int plain) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2bf(v[e]);
    st_main(reinterpret_cast<u32x4*>(p), as_u32x4(o), plain);
}

// fp16 outputs saturate at the largest finite half instead of overflowing to inf (an inf would turn the next LayerNorm
// row into NaNs); real BERT activations stay orders of magnitude below it.
__device__ __forceinline__ _Float16 f2h(float v) { return (_Float16)fminf(fmaxf(v, -65504.f), 65504.f); }
__device__ __forceinline__ void store8_f16(void* p, const float (&v)[8], int plain = 0) {
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = f2h(v[e]);
    st_main(reinterpret_cast<u32x4*>(p), __builtin_bit_cast(u32x4, o), plain);
}

// Block -> output tile.  Blocks b and b+8 share an XCD (round-robin dispatch); each XCD has a private 4 MiB L2.
//  * few column tiles: every XCD takes a contiguous run of row-major tiles (its A row panels + all of B stay in L2);
//  * many column tiles (N >= 1536): the 8 XCDs form a 4 x 2 grid over the tile matrix, so an XCD touches nbm/4 A
//    panels and nbn/2 B panels instead of nbm/8 and ALL nbn (B alone would overflow its L2 and be re-fetched from
//    the Infinity Cache for every tile row: rocprofv3 FETCH_SIZE was 3-4x the algorithmic bytes).
// Placement only affects speed: every tile is produced exactly once for any dispatch order.
__device__ __forceinline__ void tile_origin(int bid, int nb, int nbm, int nbn, int& m0, int& n0, int bn = BN) {
    const int xcd = bid & 7, li = bid >> 3;
    if (nbn >= 12 && (nb & 7) == 0 && (nbm & 3) == 0 && (nbn & 1) == 0) {
        const int sm = nbm >> 2, sn = nbn >> 1;
        const int xi = xcd >> 1, xj = xcd & 1;
        m0 = (xi * sm + li / sn) * BM;
        n0 = (xj * sn + li % sn) * bn;
        return;
    }
    const int qn = nb >> 3, rn = nb & 7;
    const int sw = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + li;
    if (nbn > nbm) {  // runs go along the SHORTER side, so an XCD's run is a squarish patch (fewer operand panels)
        m0 = (sw % nbm) * BM;
        n0 = (sw / nbm) * bn;
        return;
    }
    m0 = (sw / nbn) * BM;
    n0 = (sw % nbn) * bn;
}

// Second half of the LDS-staged epilogue: thread t finishes 8 consecutive columns (c8 = t & 15) of rows
// (t >> 4) + RSTEP*i; 16 lanes cover a whole 128-column row -> 16-byte row-contiguous global accesses.
// WIDE: 0 = a 128-row tile of the 8-wave kernels; 1 / 2 = one 128-row pass of gemm_w3_kernel's 192- / 128-column tile (the
// staged rows are 32-row slabs of four 64-row wave tiles; 192 columns use the off_cw image)
// WIRE: the instance may be asked for the data-parallel wire copy of an f32 output (C3 / c3_only): weight-gradient (TN)
// instances only -- compiled out everywhere else (as a run-time test in every instance it cost the 12-wave kernels, which run
// at their register cap, 5 % and showed up in kernels that never see a wire copy)
template <int RSTEP, int NC8 = 16, int WIDE = 0, bool WIRE = false>
__device__ __forceinline__ void epilogue_rows(const GemmArgs& g, const char* smem, int m0, int n0, int tid) {
    constexpr int ROWS = 128;   // rows of the staged tile
    // NC8 = 8-column groups per tile row: 16 (128-wide tile), 12 (96-wide: 384 of 512 threads) or 24 (192-wide, 768 threads)
    if (tid >= RSTEP * NC8) return;
    const int c8 = NC8 == 16 ? (tid & 15) : tid % NC8;   // 8-column group of the row
    const int rb = NC8 == 16 ? (tid >> 4) : tid / NC8;
    const int n = n0 + 8 * c8;
    float bias[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = 0.f;
    if (g.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias + n), b1 = *reinterpret_cast<const f32x4*>(g.bias + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias[e] = b0[e]; bias[4 + e] = b1[e]; }
    }
    if (g.bias2) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(g.bias2 + n), b1 = *reinterpret_cast<const f32x4*>(g.bias2 + n + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias[e] += b0[e]; bias[4 + e] += b1[e]; }
    }
    // The epilogue operand (fan-in gradient, GELU' input, gate factor) of ALL of this thread's rows is requested here, before
    // the first staged row is read: inside the row loop (behind the switch on the epilogue kind) every row's load waited for
    // the previous row's store to issue -- 128 / RSTEP global-load latencies in series per thread, twice that in the two
    // passes of the 12-wave kernel.
    constexpr int NROW = ROWS / RSTEP;
    const bool use_aux = g.epi == ICKA_EPI_DGELU || g.epi == ICKA_EPI_ADD || g.epi == ICKA_EPI_GATE || g.epi == ICKA_EPI_ADD_RELU;
    u32x4 araw[NROW];
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
        const int row = rb + RSTEP * i;
        const int m = (WIDE == 1 || WIDE == 2) ? m0 + ((row >> 5) << 6) + (row & 31) : m0 + row;
        araw[i] = u32x4{0u, 0u, 0u, 0u};
        if (use_aux) araw[i] = ld_once(reinterpret_cast<const u32x4*>(g.aux + (int64_t)m * g.ldaux + n));
    }
    // (the 12-wave kernel runs at a 168-register cap with accumulators of the other pass alive: keep its row loop rolled)
    constexpr int UNR = (WIDE == 1 || WIDE == 2) ? 1 : ROWS / RSTEP;
#pragma unroll UNR
    for (int i = 0; i < ROWS / RSTEP; ++i) {
        const int row = rb + RSTEP * i;
        // WIDE: the staged tile holds 32-row slabs of four 64-row wave tiles (gemm_w3_kernel): slab q -> rows 64 q + 0..31
        const int m = (WIDE == 1 || WIDE == 2) ? m0 + ((row >> 5) << 6) + (row & 31) : m0 + row;
        const u32x4 acur = araw[0];   // (constant indices: the queue stays in registers when the loop is rolled)
#pragma unroll
        for (int q = 0; q + 1 < NROW; ++q) araw[q] = araw[q + 1];
        const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + (WIDE == 1 ? off_cw(row, 2 * c8) : off_c(row, 2 * c8)));
        const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + (WIDE == 1 ? off_cw(row, 2 * c8 + 1) : off_c(row, 2 * c8 + 1)));
        float v[8] = {lo[0] + bias[0], lo[1] + bias[1], lo[2] + bias[2], lo[3] + bias[3],
                      hi[0] + bias[4], hi[1] + bias[5], hi[2] + bias[6], hi[3] + bias[7]};
        float a[8];
        switch (g.epi) {
            case ICKA_EPI_GELU:
                store8_bf16(g.C2 + (int64_t)m * g.ldc2 + n, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
                break;
            case ICKA_EPI_DGELU:
                cvt8_aux(acur, g.aux_f16, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= dgelu_f(a[e]);
                break;
            case ICKA_EPI_ADD:
                cvt8_aux(acur, g.aux_f16, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += a[e];
                break;
            case ICKA_EPI_GATE:
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = sigmoid_f(v[e]);
                if (g.C2) store8_bf16(g.C2 + (int64_t)m * g.ldc2 + n, v);
                cvt8_aux(acur, g.aux_f16, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= a[e];
                break;
            case ICKA_EPI_TANH:
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
                break;
            case ICKA_EPI_ADD_RELU:
                cvt8_aux(acur, g.aux_f16, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + a[e], 0.f);
                break;
            case ICKA_EPI_RELU:
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                break;
            default: break;
        }
        if (g.c_f32) {
            float* p = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n;
            f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
            if (g.beta != 0.f) {
                o0 += g.beta * ld_once(reinterpret_cast<const f32x4*>(p));
                o1 += g.beta * ld_once(reinterpret_cast<const f32x4*>(p + 4));
            }
            if (!WIRE || !g.c3_only) {
                st_out(reinterpret_cast<f32x4*>(p), o0);
                st_out(reinterpret_cast<f32x4*>(p + 4), o1);
            }
            if constexpr (WIRE) {
                if (g.C3) {
                    const float w[8] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3]};
                    store8_bf16(g.C3 + (int64_t)m * g.ldc3 + n, w);
                }
            }
        } else if (g.c_f16) {   // fp16 main output (+ optional bf16 copy); beta is rejected on the host
            store8_f16(reinterpret_cast<_Float16*>(g.C) + (int64_t)m * g.ldc + n, v, g.plain);
            if (g.C3) store8_bf16(g.C3 + (int64_t)m * g.ldc3 + n, v);
        } else {
            bf16_t* p = reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc + n;
            if (g.beta != 0.f) {
                load8_bf16(p, a);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += g.beta * a[e];
            }
            store8_bf16_main(p, v, g.plain);
        }
    }
}

template <bool A_KM, bool B_KM, int NBUF, int ABL>
__device__ __forceinline__ void gemm_dma_body(const GemmArgs& g, char* smem, const int bid, const int nb) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(char, smem)));
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    int m0, n0;
    tile_origin(bid, nb, g.M / BM, g.N / BN, m0, n0);

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
    const int k1t = g.K1 > 0 ? g.K1 / BK : -1;  // k-tile at which the reduction switches to (A2, B2)
    const bf16_t* pa[4];
    const bf16_t* pb[4];
    dma_init<A_KM>(pa, g.A, g.lda, m0, wave, lane);
    dma_init<B_KM>(pb, g.B, g.ldb, n0, wave, lane);
    int64_t sa = A_KM ? (int64_t)BK * g.lda : BK, sb = B_KM ? (int64_t)BK * g.ldb : BK;

#define ICKA_STAGE(KT, BUF)                                             \
    do {                                                                \
        if ((KT) == k1t) {                                              \
            dma_init<A_KM>(pa, g.A2, g.lda2, m0, wave, lane);           \
            dma_init<B_KM>(pb, g.B2, g.ldb2, n0, wave, lane);           \
            sa = A_KM ? (int64_t)BK * g.lda2 : BK;                      \
            sb = B_KM ? (int64_t)BK * g.ldb2 : BK;                      \
        }                                                               \
        dma_issue(pa, sa, lds0 + (BUF) + wave * 1024);                  \
        dma_issue(pb, sb, lds0 + (BUF) + TILE_BYTES + wave * 1024);     \
    } while (0)

    // LDS ring of NBUF stages, prefetch distance NBUF-1 k-tiles.  Each wave waits for ITS pieces of tile kt with a
    // counted vmcnt (the 8 DMA of tile kt+1 may stay in flight), then one raw s_barrier makes every wave's pieces
    // visible and also proves all waves finished reading the buffer that the next DMA overwrites.
    if (NBUF < 4) {
        // ---- v2 loop: ring of NBUF stages, fragments read right after the barrier -------------------------------
#pragma unroll
        for (int t = 0; t < NBUF - 1; ++t)
            if (t < nk) ICKA_STAGE(t, t * 2 * TILE_BYTES);
        int cur = 0;
#ifdef ICKA_GEMM_STAMP
        unsigned long long seg[5] = {0, 0, 0, 0, 0}, tA, tB;
        const unsigned long long real0 = __builtin_amdgcn_s_memrealtime(), cyc0 = __builtin_amdgcn_s_memtime();
#define STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#else
#define STAMP(v)
#endif
        for (int kt = 0; kt < nk; ++kt) {
#ifdef ICKA_GEMM_STAMP
            STAMP(tA);
#endif
            // tiles kt+1 .. min(kt+NBUF-2, nk-1) may stay in flight (8 DMA each per wave)
            int ahead = nk - 1 - kt;
            ahead = ahead > NBUF - 2 ? NBUF - 2 : ahead;
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ICKA_GEMM_STAMP
            STAMP(tB); seg[0] += tB - tA; tA = tB;
#endif
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#ifdef ICKA_GEMM_STAMP
            STAMP(tB); seg[1] += tB - tA; tA = tB;
#endif
            if (kt + NBUF - 1 < nk && ABL != 2) {
                int nx = cur + NBUF - 1;
                nx = nx >= NBUF ? nx - NBUF : nx;
                ICKA_STAGE(kt + NBUF - 1, nx * 2 * TILE_BYTES);
            }
#ifdef ICKA_GEMM_STAMP
            STAMP(tB); seg[2] += tB - tA; tA = tB;
#endif
            const char* sA = smem + cur * 2 * TILE_BYTES;
            const char* sB = sA + TILE_BYTES;
            if (ABL != 1)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = read_frag<A_KM>(sA, wr + 16 * t, ks, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[t] = read_frag<B_KM>(sB, wc + 16 * t, ks, lane);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(fb[ni], fa[mi], acc[mi][ni]);
            }
            cur = cur + 1 == NBUF ? 0 : cur + 1;
#ifdef ICKA_GEMM_STAMP
            __builtin_amdgcn_sched_barrier(0);
            STAMP(tB); seg[3] += tB - tA;
#endif
        }
#ifdef ICKA_GEMM_STAMP
        if (g.stamp && (tid & 63) == 0 && wave == 0) {
            unsigned long long* o = g.stamp + (size_t)bid * 8;
            o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2]; o[3] = seg[3];
            o[4] = __builtin_amdgcn_s_memtime() - cyc0;
            o[5] = __builtin_amdgcn_s_memrealtime() - real0;
            o[6] = nk;
        }
#endif
    } else {
        // ---- v3 loop (ring of 4, one block per CU = one wave per SIMD): software-pipelined FRAGMENTS.  The barrier
        // sits between the two 16-MFMA halves of a k-tile: the second half's fragments (ks=1 of tile kt) are read
        // under the first half's MFMAs, and the NEXT tile's first fragments (ks=0 of tile kt+1, made visible by
        // this iteration's barrier) plus the DMA issue of tile kt+3 go under the second half's MFMAs, so neither
        // the LDS latency nor the DMA issue sequence is exposed at one wave per SIMD.
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (t < nk) ICKA_STAGE(t, t * 2 * TILE_BYTES);
        if (nk >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tiles 0,1 landed; tile 2 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fa0[t] = read_frag<A_KM>(smem, wr + 16 * t, 0, lane);
#pragma unroll
        for (int t = 0; t < 4; ++t) fb0[t] = read_frag<B_KM>(smem + TILE_BYTES, wc + 16 * t, 0, lane);
        int cur = 0;
#ifdef ICKA_GEMM_STAMP
        unsigned long long cseg[2] = {0, 0}, cA, cB;
        const unsigned long long ccyc0 = __builtin_amdgcn_s_memtime();
#endif
        for (int kt = 0; kt < nk; ++kt) {
#ifdef ICKA_GEMM_STAMP
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cA)::"memory");
#endif
            const char* sA = smem + cur * 2 * TILE_BYTES;
            const char* sB = sA + TILE_BYTES;
#pragma unroll
            for (int t = 0; t < 4; ++t) fa1[t] = read_frag<A_KM>(sA, wr + 16 * t, 1, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) fb1[t] = read_frag<B_KM>(sB, wc + 16 * t, 1, lane);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(fb0[ni], fa0[mi], acc[mi][ni]);
            const int nxt = cur == 3 ? 0 : cur + 1;
            if (kt + 1 < nk) {
                // tile kt+1 must have landed (own pieces); tile kt+2 (if any) may stay in flight
                if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (kt + 3 < nk) {
                    const int st = cur == 0 ? 3 : cur - 1;   // buffer of tile kt-1 == (kt+3) % 4
                    ICKA_STAGE(kt + 3, st * 2 * TILE_BYTES);
                }
                const char* nA = smem + nxt * 2 * TILE_BYTES;
#pragma unroll
                for (int t = 0; t < 4; ++t) fa0[t] = read_frag<A_KM>(nA, wr + 16 * t, 0, lane);
#pragma unroll
                for (int t = 0; t < 4; ++t) fb0[t] = read_frag<B_KM>(nA + TILE_BYTES, wc + 16 * t, 0, lane);
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(fb1[ni], fa1[mi], acc[mi][ni]);
            cur = nxt;
        }
    }
#undef ICKA_STAGE
    __syncthreads();  // every wave is done with the operand ring before it is reused as the C tile

    // ---- epilogue through LDS (all waves are past the last barrier: the operand buffers are dead)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int row = wr + 16 * mi + (lane & 15);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int ch = (wc >> 2) + 4 * ni + (lane >> 4);
            *reinterpret_cast<f32x4*>(smem + off_c(row, ch)) = acc[mi][ni] * g.alpha;
        }
    }
    __syncthreads();
    epilogue_rows<16, 16, 0, A_KM>(g, smem, m0, n0, tid);
}

__device__ __forceinline__ bool g_direct_epilogue(const GemmArgs& g) {
    // f32 outputs only: 16 B per lane.  bf16 outputs (8 B per lane, 32-byte row pieces) measured 5 % slower than the
    // staged 16-byte row-contiguous stores (qkv projection, profiles/README.md)
    return g.direct && g.c_f32 && g.epi == ICKA_EPI_NONE && g.beta == 0.f;
}

// =====================================================================================================================
// Warp-specialised fast path (512 threads): waves 0-3 COMPUTE (LDS fragment reads + MFMA, 64x64 each), waves 4-7 LOAD
// (they only issue the LDS-DMA of the ring and wait for it).  In-kernel stamps of the 4-wave kernel showed a wave
// spending ~500 cycles per k-tile issuing its 8 global_load_lds (~60 cycles each) in series with its 512 cycles of
// MFMA; a loader wave co-resident on the same SIMD hides that issue time under the compute wave's matrix work.
// One s_barrier per k-tile joins both roles: loaders arrive after their counted vmcnt (tile kt landed), compute waves
// after finishing tile kt-1, so the barrier both publishes tile kt and frees the buffer of tile kt-1 for re-staging.
// LNF (gemm_ln_kernel): the tile's plain f32 output is stored WRITE-THROUGH (sc1) and the body returns instead of retiring its
// waves -- the kernel then hands the stripe's rows over to its LayerNorm phase inside the same launch.
template <bool A_KM, bool B_KM, int NBUF, int ABL = 0, int DIST = 2, int BNT = 128, bool F16 = false, bool CONV = false, bool LNF = false>
__device__ __forceinline__ void gemm_ws_body(const GemmArgs& g, char* smem, const int bid, const int nb) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(char, smem)));
    int m0, n0;
    // BNT = tile width: 128, or 96 when that fills the 256 CUs better (N = 768 -> 256 tiles instead of 192)
    constexpr int NTN = BNT / 32;                  // 16-column MFMA tiles per compute wave (4 or 3)
    constexpr int NJB = B_KM ? 4 : BNT / 32;       // LDS-DMA pieces of the B tile per loader wave
    constexpr int ND = 4 + NJB;                    // DMA instructions per k-tile per loader wave
    static_assert(BNT == 128 || BNT == 96 || BNT == 64, "tile width");
    tile_origin(bid, nb, g.M / BM, g.N / BNT, m0, n0, BNT);
    const int nk = g.K / BK;
    const int wr = ((wave & 3) >> 1) * 64, wc = (wave & 1) * (BNT / 2);
#ifdef ICKA_GEMM_STAMP
    const unsigned long long ph0 = __builtin_amdgcn_s_memtime();
    unsigned long long ph1 = 0, ph2 = 0;
#endif

    f32x4 acc[4][NTN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Bias gradient for free: in the blocks of the first tile column the compute waves that own columns 0..63 also
    // multiply the A fragments (dY, rows = output features) by an all-ones operand: D[i][j] = sum_k A[j][k], i.e. the
    // column sums of dY over the tokens, 4 extra MFMAs per 16 (only in 1/nbn of the blocks).
    const bool do_cs = A_KM && g.colsum != nullptr && n0 == 0 && wave < 4 && wc == 0;
    f32x4 cs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f), f2bf(1.f)};

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------- loader waves
        const int lw = wave - 4;
        const int k1t = g.K1 > 0 ? g.K1 / BK : -1;
        const bf16_t* pa[4];
        const bf16_t* pb[4];
        // CONV: the A tile is gathered from the NHWC activation.  Per piece (8 tile rows) this lane serves one output pixel:
        // the pointer to its centre input pixel (+ this lane's 16-byte chunk) and a 9-bit mask of the taps that lie inside
        // the image; a k-tile is 64 channels of ONE tap (cvC % 64 == 0), so its source is centre + a wave-uniform offset,
        // or the zero page.
        const bf16_t* ctr[4];
        uint32_t vmask[4] = {0u, 0u, 0u, 0u};
        const bf16_t* zsrc = nullptr;
        int cv_tap = 0, cv_cb = 0;
        if constexpr (CONV) {
            static_assert(!A_KM && !B_KM, "implicit convolution: NT only");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = 8 * (lw + 4 * j) + (lane >> 3);
                const int lc = (lane & 7) ^ ((row >> 1) & 7);
                const int m = m0 + row;
                const int mm = m < g.cvRows ? m : 0;
                const int hw = g.cvHo * g.cvWo;
                const int b = mm / hw, r = mm - b * hw, oy = r / g.cvWo, ox = r - oy * g.cvWo;
                const int y0 = oy * g.cvS, x0 = ox * g.cvS;
                ctr[j] = g.A + ((int64_t)(b * g.cvH + y0) * g.cvW + x0) * g.cvC + 8 * lc;
                uint32_t vm = 0u;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int y = y0 + t / 3 - 1, x = x0 + t % 3 - 1;
                    vm |= (m < g.cvRows && y >= 0 && y < g.cvH && x >= 0 && x < g.cvW) ? (1u << t) : 0u;
                }
                vmask[j] = vm;
                if (j == 0) zsrc = g.cvZero + 8 * (lane & 7);
            }
        } else {
            dma_init<A_KM>(pa, g.A, g.lda, m0, lw, lane);
        }
        dma_init<B_KM, BNT>(pb, g.B, g.ldb, n0, lw, lane);
        int64_t sa = CONV ? 0 : (A_KM ? (int64_t)BK * g.lda : BK), sb = B_KM ? (int64_t)BK * g.ldb : BK;
#define ICKA_WS_STAGE(KT, BUF)                                          \
    do {                                                                \
        if ((KT) == k1t) {                                              \
            dma_init<A_KM>(pa, g.A2, g.lda2, m0, lw, lane);             \
            dma_init<B_KM, BNT>(pb, g.B2, g.ldb2, n0, lw, lane);        \
            sa = A_KM ? (int64_t)BK * g.lda2 : BK;                      \
            sb = B_KM ? (int64_t)BK * g.ldb2 : BK;                      \
        }                                                               \
        if constexpr (CONV) {   /* k-tiles are staged in order: (tap, channel block) advance as counters */ \
            const int64_t off_ = (int64_t)((cv_tap / 3 - 1) * g.cvW + (cv_tap % 3 - 1)) * g.cvC + cv_cb * 64; \
            _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)            \
                pa[j_] = ((vmask[j_] >> cv_tap) & 1u) ? ctr[j_] + off_ : zsrc; \
            if (++cv_cb == g.cvC / 64) { cv_cb = 0; ++cv_tap; }         \
        }                                                               \
        if (ABL != 2) {                                                 \
            dma_issue(pa, sa, lds0 + (BUF) + lw * 1024);                \
            dma_issue<NJB>(pb, sb, lds0 + (BUF) + TILE_BYTES + lw * 1024); \
        }                                                               \
    } while (0)
#pragma unroll
        for (int t = 0; t < NBUF - 1; ++t)
            if (t < nk) ICKA_WS_STAGE(t, t * 2 * TILE_BYTES);
        int cur = 0;
#ifdef ICKA_GEMM_STAMP
        unsigned long long seg[4] = {0, 0, 0, 0}, tA, tB;
        const unsigned long long real0 = __builtin_amdgcn_s_memrealtime(), cyc0 = __builtin_amdgcn_s_memtime();
#define WSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory")
#endif
        for (int kt = 0; kt < nk; ++kt) {
#ifdef ICKA_GEMM_STAMP
            WSTAMP(tA);
#endif
            int ahead = nk - 1 - kt;
            ahead = ahead > NBUF - 2 ? NBUF - 2 : ahead;
            // tile kt has landed once at most `ahead` younger tiles (ND DMA instructions each) are still in flight
            static_assert(NBUF <= 5 && ND * (NBUF - 2) <= 63, "vmcnt range");
            switch (ahead) {
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<ND>(); break;
                case 2: wait_vmcnt<2 * ND>(); break;
                default: wait_vmcnt<(NBUF > 4 ? 3 : 0) * ND>(); break;
            }
#ifdef ICKA_GEMM_STAMP
            WSTAMP(tB); seg[0] += tB - tA; tA = tB;
#endif
            if (ABL != 3) __builtin_amdgcn_s_barrier();   // (ABL 3, diagnostic: both roles free-running, garbage results)
#ifdef ICKA_GEMM_STAMP
            WSTAMP(tB); seg[1] += tB - tA; tA = tB;
#endif
            if (kt + NBUF - 1 < nk) {
                int nx = cur + NBUF - 1;
                nx = nx >= NBUF ? nx - NBUF : nx;
                ICKA_WS_STAGE(kt + NBUF - 1, nx * 2 * TILE_BYTES);
            }
            cur = cur + 1 == NBUF ? 0 : cur + 1;
#ifdef ICKA_GEMM_STAMP
            WSTAMP(tB); seg[2] += tB - tA;
#endif
        }
#ifdef ICKA_GEMM_STAMP
        if (g.stamp && lane == 0 && wave == 4) {
            unsigned long long* o = g.stamp + (size_t)bid * 16;
            o[0] = seg[0]; o[1] = seg[1]; o[2] = seg[2];
            o[4] = __builtin_amdgcn_s_memtime() - cyc0;
            o[5] = __builtin_amdgcn_s_memrealtime() - real0;
            o[6] = nk;
        }
#endif
#undef ICKA_WS_STAGE
    } else {
        // ------------------------------------------------------------------------------------------ compute waves
        // Fragments are software-pipelined in registers with a prefetch distance of TWO 16-MFMA halves: while tile
        // kt is multiplied out of one register set (P), both halves of tile kt+1 are read into the other (Q).  Measured
        // (tools/probe/mfma_probe): LDS read latency at one wave per SIMD is hundreds of cycles once LDS-DMA writes
        // share the LDS pipe, far more than one half (272 cycles) covers.  Barrier kt+1 (tile kt+1 published, tile
        // kt fully in registers -> its buffer may be re-staged) is taken at the TOP of iteration kt.
        if constexpr (DIST == 1) {
            // two blocks per CU (ring of 2, <= 128 VGPRs): fragments are read per 32-deep step right before their
            // MFMAs; the co-resident block's waves cover the LDS latency and this block's epilogue/prologue
            int cur = 0;
            for (int kt = 0; kt < nk; ++kt) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const char* sA = smem + cur * 2 * TILE_BYTES;
                const char* sB = sA + TILE_BYTES;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 fa[4], fb[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) fa[t] = read_frag<A_KM>(sA, wr + 16 * t, ks, lane);
#pragma unroll
                    for (int t = 0; t < NTN; ++t) fb[t] = read_frag<B_KM>(sB, wc + 16 * t, ks, lane);
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NTN; ++ni) acc[mi][ni] = mfma16t<F16>(fb[ni], fa[mi], acc[mi][ni]);
                    if (do_cs) {
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) cs[mi] = mfma16(ones, fa[mi], cs[mi]);
                    }
                }
                cur = cur + 1 == NBUF ? 0 : cur + 1;
            }
        } else {
        bf16x8 pa0[4], pb0[4], pa1[4], pb1[4], qa0[4], qb0[4], qa1[4], qb1[4];
#define ICKA_READ(FA, FB, BUFI, KS)                                                                  \
    if (ABL != 1) do {                                                                               \
        const char* b_ = smem + (BUFI) * 2 * TILE_BYTES;                                             \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) FA[t] = read_frag<A_KM>(b_, wr + 16 * t, KS, lane);              \
        _Pragma("unroll") for (int t = 0; t < NTN; ++t) FB[t] = read_frag<B_KM>(b_ + TILE_BYTES, wc + 16 * t, KS, lane); \
        __builtin_amdgcn_sched_barrier(0); /* keep the reads AHEAD of the next MFMA group (hipcc sinks them) */     \
    } while (0)
#define ICKA_MMA(FA, FB)                                                                             \
    if (ABL != 1) do {                                                                               \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                             \
            _Pragma("unroll") for (int ni = 0; ni < NTN; ++ni) acc[mi][ni] = mfma16t<F16>(FB[ni], FA[mi], acc[mi][ni]); \
        if (do_cs) { _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) cs[mi] = mfma16(ones, FA[mi], cs[mi]); }  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
// ICKA_RM: one scheduling region = fragment reads of the NEXT k-step (into the idle register set) + the MFMAs of the
// current one, with the DS reads spread between the MFMAs (sched_group_barrier: 1 MFMA, 1 DS read, ...).  Issued as a
// block in front of the MFMAs, the reads (+ their address arithmetic) left the matrix pipe idle for 150-300 cycles
// per k-step: there is one compute wave per SIMD, nothing else fills those slots (measured on the 256x128 kernel:
// 80.9 -> 64.3 us per launch).
#define ICKA_RM(RA, RB, BUFI, KS, MA, MB)                                                            \
    if (ABL != 1) do {                                                                               \
        constexpr int NR_ = 4 * (A_KM ? 2 : 1) + NTN * (B_KM ? 2 : 1), NM_ = 4 * NTN;                \
        constexpr int NP_ = NR_ < NM_ ? NR_ : NM_;                                                   \
        const char* b_ = smem + (BUFI) * 2 * TILE_BYTES;                                             \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) RA[t] = read_frag<A_KM>(b_, wr + 16 * t, KS, lane);              \
        _Pragma("unroll") for (int t = 0; t < NTN; ++t) RB[t] = read_frag<B_KM>(b_ + TILE_BYTES, wc + 16 * t, KS, lane); \
        _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                             \
            _Pragma("unroll") for (int ni = 0; ni < NTN; ++ni) acc[mi][ni] = mfma16t<F16>(MB[ni], MA[mi], acc[mi][ni]); \
        if constexpr (NR_ > NM_) __builtin_amdgcn_sched_group_barrier(0x100, NR_ - NM_, 0);          \
        _Pragma("unroll") for (int i_ = 0; i_ < NP_; ++i_) {                                         \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                       \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                       \
        }                                                                                            \
        if constexpr (NM_ > NR_) __builtin_amdgcn_sched_group_barrier(0x008, NM_ - NR_, 0);          \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (do_cs) { _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) cs[mi] = mfma16(ones, MA[mi], cs[mi]); }  \
        __builtin_amdgcn_sched_barrier(0);                                                           \
    } while (0)
#define ICKA_SYNC()                                          \
    do {                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
        if (ABL != 3) __builtin_amdgcn_s_barrier();          \
        asm volatile("" ::: "memory");                       \
    } while (0)
        if (ABL != 3) __builtin_amdgcn_s_barrier();   // barrier #0: tile 0 published
        asm volatile("" ::: "memory");
#ifdef ICKA_GEMM_STAMP
        ph1 = __builtin_amdgcn_s_memtime();
#endif
        ICKA_READ(pa0, pb0, 0, 0);
        ICKA_READ(pa1, pb1, 0, 1);
        int nxt = NBUF > 1 ? 1 : 0;     // ring slot of tile kt+1
        int kt = 0;
        for (; kt + 2 <= nk - 1; kt += 2) {
            ICKA_SYNC();                         // barrier kt+1
            ICKA_RM(qa0, qb0, nxt, 0, pa0, pb0);
            ICKA_RM(qa1, qb1, nxt, 1, pa1, pb1);
            nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
            ICKA_SYNC();                         // barrier kt+2
            ICKA_RM(pa0, pb0, nxt, 0, qa0, qb0);
            ICKA_RM(pa1, pb1, nxt, 1, qa1, qb1);
            nxt = nxt + 1 == NBUF ? 0 : nxt + 1;
        }
        // tail: kt is the next tile to multiply (in P); nk - kt is 1 or 2
        if (kt + 1 <= nk - 1) {
            ICKA_SYNC();
            ICKA_RM(qa0, qb0, nxt, 0, pa0, pb0);
            ICKA_RM(qa1, qb1, nxt, 1, pa1, pb1);
            ICKA_MMA(qa0, qb0);
            ICKA_MMA(qa1, qb1);
        } else {
            ICKA_MMA(pa0, pb0);
            ICKA_MMA(pa1, pb1);
        }
#undef ICKA_READ
#undef ICKA_MMA
#undef ICKA_RM
#undef ICKA_SYNC
        }
        // MFMA -> VALU read-after-write needs software wait states on gfx950 (8-pass MFMA: ~11).  hipcc's hazard
        // recognizer missed one across a block boundary here (<TN, 96-wide>, odd k-tile count: a v_mov of the last
        // accumulator element right behind the branch that follows the last MFMA -> one stale element per lane, found
        // by tools/gemm_tile_check.py), so the compute waves always idle 16 states before anything reads acc.
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#ifdef ICKA_GEMM_STAMP
        ph2 = __builtin_amdgcn_s_memtime();
#endif
    }
    // Plain f32 outputs (no activation / fan-in operand / accumulate: the GEMM -> LayerNorm intermediates) go straight
    // from the accumulators to HBM: a lane owns 4 consecutive columns of one row (16 B) and the 4 lane groups of an
    // MFMA tile cover 64 contiguous bytes per row; the stores of adjacent tiles merge in L2.  This skips two block barriers and
    // 128 KB of LDS traffic of the staged epilogue below, and the loader waves retire at once.
    if (g_direct_epilogue(g)) {
#ifdef ICKA_GEMM_STAMP
        if (g.stamp && lane == 0 && wave == 0) {
            unsigned long long* o = g.stamp + (size_t)bid * 16 + 11;
            o[0] = ph1 - ph0; o[1] = ph2 - ph1; o[2] = 0;
        }
#endif
        if (wave < 4) {
            __amdgpu_buffer_rsrc_t crsrc = __builtin_amdgcn_make_buffer_rsrc(g.C, 0, 0x7ffffff0, 0x00020000);   // (LNF stores)
            (void)crsrc;
            if (do_cs && lane < 16) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    float* p = g.colsum + m0 + wr + 16 * mi + lane;
                    *p = g.colsum_acc ? *p + cs[mi][0] : cs[mi][0];
                }
            }
#pragma unroll
            for (int ni = 0; ni < NTN; ++ni) {
                const int n = n0 + wc + 16 * ni + 4 * (lane >> 4);
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
                if (g.bias) b4 = *reinterpret_cast<const f32x4*>(g.bias + n);
                if (g.bias2) b4 += *reinterpret_cast<const f32x4*>(g.bias2 + n);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const int m = m0 + wr + 16 * mi + (lane & 15);
                    const f32x4 v = acc[mi][ni] * g.alpha + b4;
                    if constexpr (LNF) {
                        // write-through (sc1 raw buffer store): the rows are read by OTHER CUs of this launch (ln_row.h).  (A first
                        // version used an inline-asm global_store ... sc1: hipcc's hazard recognizer cannot see into it, and the
                        // next tile's v_pk_fma overwrote the data registers of the store in flight -- 10 % wrong elements.)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), crsrc, (int)(((int64_t)m * g.ldc + n) * 4), 0, 16);
                    } else if (g.c_f32) {
                        if (!A_KM || !g.c3_only) st_out(reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n), v);
                        if constexpr (A_KM) {   // weight-gradient instances only: the data-parallel wire copy
                            if (g.C3) st_out(reinterpret_cast<u32x2*>(g.C3 + (int64_t)m * g.ldc3 + n), pack4(v[0], v[1], v[2], v[3]));
                        }
                    } else st_main(reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(g.C) + (int64_t)m * g.ldc + n),
                                pack4(v[0], v[1], v[2], v[3]), g.plain);
                }
            }
        }
        return;
    }
    __syncthreads();  // every wave is done with the operand ring before it is reused as the C tile

    if (do_cs && lane < 16) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            float* p = g.colsum + m0 + wr + 16 * mi + lane;
            *p = g.colsum_acc ? *p + cs[mi][0] : cs[mi][0];
        }
    }
    // ---- epilogue through LDS: compute waves deposit their accumulators, all 8 waves finish rows
    if (wave < 4) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int row = wr + 16 * mi + (lane & 15);
#pragma unroll
            for (int ni = 0; ni < NTN; ++ni) {
                const int ch = (wc >> 2) + 4 * ni + (lane >> 4);
                *reinterpret_cast<f32x4*>(smem + off_c(row, ch)) = acc[mi][ni] * g.alpha;
            }
        }
    }
    __syncthreads();
    epilogue_rows<32, BNT / 8, 0, A_KM>(g, smem, m0, n0, tid);
#ifdef ICKA_GEMM_STAMP
    if (g.stamp && lane == 0 && wave == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* o = g.stamp + (size_t)bid * 16 + 11;
        o[0] = ph1 - ph0; o[1] = ph2 - ph1; o[2] = __builtin_amdgcn_s_memtime() - ph2;
        o[3] = ph0; o[4] = __builtin_amdgcn_s_memtime();
    }
#endif
}

template <bool A_KM, bool B_KM, int NBUF, int ABL = 0, int BNT = 128, bool F16 = false, bool CONV = false>
__global__ __launch_bounds__(512) void gemm_ws_kernel(const GemmArgs gp) {
    const GemmArgs g = gp;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE_BYTES];
    gemm_ws_body<A_KM, B_KM, NBUF, ABL, 2, BNT, F16, CONV>(g, smem, blockIdx.x, gridDim.x);
}

// =====================================================================================================================
// dense -> bias + dropout + residual -> LayerNorm in ONE launch (BertSelfOutput.forward Cross_Modal_Interaction_Module.py:
// 561-565, BertOutput.forward :532-536): the 128 x BNT NT kernel above for shapes whose tile grid is one block per CU with
// EIGHT column tiles per 128-row stripe (N = 768 with 96-wide tiles, N = 1024 with 128-wide ones; M / 128 stripes, at most one
// block per CU).  Where M / 128 is a multiple of 8, tile_origin puts the 8 blocks of a stripe on one XCD (blocks b, b + 8, ...
// share an XCD under the observed round-robin dispatch) -- a speed property only, nothing below depends on it: with other stripe
// counts (the reference's test loop at batch 4: M = 512) a stripe spreads over two or three XCDs and the hand-off still holds.
//   phase 1  the tile's f32 output is stored write-through (sc1) into the usual GEMM -> LayerNorm intermediate;
//   seam     every wave waits for its stores (s_waitcnt vmcnt(0)), the block barrier joins them, ONE lane adds 1 to the stripe's
//            arrival counter (agent scope) and polls it (sc1 load, bounded) until all 8 blocks of the stripe have arrived;
//            measured 0.6 us (max 0.8) on an idle chip against 4.8 us first-block-start -> last-block-end for the same two
//            phases as two launches (tools/probe/xcd_seam_probe.hip, profiles/r05_xcd_seam_probe.txt);
//   phase 2  block j of the stripe finishes rows 16 j .. 16 j + 15 of it -- all 8 waves, two rows each, through the SAME row body
//            as the stand-alone kernel (ln_row.h: bitwise the same y / twin / xhat / rstd), reading the intermediate with
//            L1-bypassing (sc1) loads.
// Why the round-2 attempt at this fusion lost (profiles/r02_gemm_ln_fusion.txt: +6 - 8 us per site) and this form differs: there
// a thread kept 24 columns of a row and the 8 blocks exchanged partial (sum, M2) statistics through four dependent agent-scope
// round trips, then wrote y / twin / xhat as 48 - 96-byte pieces; here the seam is ONE counter and the LayerNorm phase moves whole
// rows, 16 bytes per lane, like the row kernel.  The counters reset themselves: the last of the 8 blocks to LEAVE the wait
// zeroes both words (by then every block of the stripe has seen the count), so a launch leaves the workspace as it found it
// and the same words serve every fused launch of a stream (launches of one stream do not overlap).
// A wait that gives up (a block of the stripe never became resident: grids of more blocks than the device's CUs minus the
// caller's reserve, icka_lstm_set_reserved_cus, are refused on the host) stores 1 into the error word -- host-mapped memory the
// host polls without a device synchronisation -- and the block turns the rows it finished into NaN: never a hang, never a
// silent wrong result.
struct GemmLnArgs {
    GemmArgs g;
    LnFwdArgs ln;
    unsigned int* sync;      // [stripes][32] words: [0] arrivals, [16] departures (one 64-byte line each)
    unsigned int* err;       // error word (host-mapped memory when the caller wants to poll it without a device sync)
    int polls;
    int test_drop;           // test hook: this block never arrives (its stripe's waits give up); -1 = off
};
template <int BNT, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_ln_kernel(const GemmLnArgs p) {
    const GemmArgs g = p.g;
    __shared__ __attribute__((aligned(16))) char smem[3 * 2 * TILE_BYTES];
    __shared__ int s_gave_up;
    int m0, n0;
    tile_origin(blockIdx.x, gridDim.x, g.M / BM, g.N / BNT, m0, n0, BNT);
    const int nbn = g.N / BNT;                             // = 8 (host)
    LnFwdArgs a = p.ln;
    a.drop = drop_resolve(a.drop);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int ROWS_PER_BLOCK = BM / 8, ROWS_PER_WAVE = ROWS_PER_BLOCK / 8;
    const int r0 = m0 + (n0 / BNT) * ROWS_PER_BLOCK + wave * ROWS_PER_WAVE;
    // the residual rows this wave will finish do not depend on the GEMM: requested now, they arrive under the k-loop instead of
    // in the tail, where all 256 blocks would ask HBM for them at the same moment (32 registers per lane, held across the loop)
    float rr[ROWS_PER_WAVE][2][8];
    ln_prefetch_res<2, ROWS_PER_WAVE>(a, r0, lane, rr);
    gemm_ws_body<false, false, 3, 0, 2, BNT, F16, false, true>(g, smem, blockIdx.x, gridDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int* arrive = p.sync + (size_t)(m0 / BM) * 32;
        unsigned int* depart = arrive + 16;
        if ((int)blockIdx.x != p.test_drop) __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = false;
        for (int i = 0; i < p.polls; ++i) {
            if (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)nbn) { ok = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        s_gave_up = ok ? 0 : 1;
        if (!ok) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the last block to LEAVE the wait zeroes the stripe's words (every block of the stripe has stopped polling by then --
        // it saw the full count or gave up): the workspace is left as it was found, also after a failed launch
        if (__hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nbn - 1)) {
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    ln_fwd_rows_handoff<2, ROWS_PER_WAVE>(a, r0, lane, rr);
    if (s_gave_up) {
        // a failed launch never passes for a result: the rows this block finished from an incomplete stripe become NaN (y is the
        // operand of everything downstream: the loss and every gradient of the step are NaN), and the host raises at its next
        // touch-point (kernels.gemm_ln_check_error)
        const bf16_t qnan = f2bf(__builtin_nanf(""));
        for (int r = 0; r < ROWS_PER_WAVE; ++r)
            for (int c = lane; c < a.H; c += 64) a.y[(int64_t)(r0 + r) * a.ldy + c] = qnan;
    }
}

// Two co-resident blocks per CU (64 KiB ring of 2 each, 4 waves per SIMD -> <= 128 VGPRs): for grids of several
// tiles per CU one block's prologue (first DMA latency, ~2.8k cycles) and epilogue (~5.5k) overlap the other
// block's main loop (stamps: at K = 768 they are 40 % of a tile's time).
template <bool A_KM, bool B_KM, int BNT = 128, bool F16 = false>
__global__ __launch_bounds__(512, 4) void gemm_ws2_kernel(const GemmArgs gp) {
    const GemmArgs g = gp;
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_BYTES];
    gemm_ws_body<A_KM, B_KM, 2, 0, 1, BNT, F16>(g, smem, blockIdx.x, gridDim.x);
}

template <bool A_KM, bool B_KM, int NBUF, int ABL>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const GemmArgs gp) {
    const GemmArgs g = gp;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE_BYTES];
    gemm_dma_body<A_KM, B_KM, NBUF, ABL>(g, smem, blockIdx.x, gridDim.x);
}

// Several independent GEMMs of one layout in ONE launch (the four weight-gradient GEMMs of a layer: 36..144 tiles
// each, 432 together): the chip is filled once instead of four partially filled launches.
constexpr int MAX_GROUP = 4;
struct GroupArgs {
    GemmArgs p[MAX_GROUP];
    int start[MAX_GROUP + 1];  // first block of each problem; start[n..] = total
};
template <bool A_KM, bool B_KM, int NBUF>
__global__ __launch_bounds__(256) void gemm_dma_group_kernel(const GroupArgs ga) {
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE_BYTES];
    const int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) pi += bid >= ga.start[i] ? 1 : 0;
    const GemmArgs g = ga.p[pi];
    gemm_dma_body<A_KM, B_KM, NBUF, 0>(g, smem, bid - ga.start[pi], ga.start[pi + 1] - ga.start[pi]);
}


// ---- launch-heuristic overrides.  NO mutable process state (launchers are re-entrant and do not rely on
// hidden state; the autograd thread and a second model see exactly what the first one does): kTuneEnv is the ICKA_TUNE_GEMM_*
// environment read ONCE when the library is loaded (same-box A/B runs: tools/ab_env.sh), and a single launch may override fields
// through icka_gemm_desc.tune (tests of the alternative kernels, tools/gemm_*.py).  Results never depend on any of it.
struct Tune {
    int abl;      // diagnostic builds (-DICKA_GEMM_ABLATE): 1 = skip MFMA + LDS reads, 2 = skip the LDS-DMA staging (wrong results)
    int ws;       // 1: warp-specialised fast path (two blocks per CU for large grids), 0: single-role kernel, 2: force two blocks, 3: never two
    int direct;   // 1: plain outputs are stored straight from the accumulators (0: always through the LDS C tile)
    int w3;       // 1: 256x192 / 256x128 tiles (12-wave kernel) for wide / short-K outputs
    int bn;       // tile width of the warp-specialised path: 0 = heuristic, 128 / 96 forced
    int nbuf;     // LDS ring depth of the fast path: 0 = per-shape heuristic, or forced 2 .. 5
    int w3grid;   // XCD cut of the 12-wave kernel's tile grid: 0 = per shape, 8 / 4 / 2 / 1 = force pm (if it divides the tile grid)
    int big;      // grouped TN launches: 2 = 256x128 tiles / 12 waves, 1 = 8 waves, 0 = 128x128 group kernel
};
static int tune_env_int(const char* name, int dflt, int lo, int hi) {
    const char* e = getenv(name);
    if (!e || !*e) return dflt;
    const int v = atoi(e);
    return v < lo || v > hi ? dflt : v;
}
static const Tune kTuneEnv = [] {
    Tune t;
    t.abl = tune_env_int("ICKA_TUNE_GEMM_ABLATION", 0, 0, 3);
    t.ws = tune_env_int("ICKA_TUNE_GEMM_WARP_SPECIALIZED", 1, 0, 3);
    t.direct = tune_env_int("ICKA_TUNE_GEMM_DIRECT_EPILOGUE", 1, 0, 1);
    t.w3 = tune_env_int("ICKA_TUNE_GEMM_WIDE_TILES", 1, 0, 1);
    t.bn = tune_env_int("ICKA_TUNE_GEMM_TILE_N", 0, 0, 128);
    if (t.bn != 0 && t.bn != 96 && t.bn != 128) t.bn = 0;
    t.nbuf = tune_env_int("ICKA_TUNE_GEMM_RING", 0, 0, 5);
    if (t.nbuf == 1) t.nbuf = 0;
    t.w3grid = tune_env_int("ICKA_TUNE_GEMM_W3_GRID", 0, 0, 8);
    if (t.w3grid != 0 && t.w3grid != 1 && t.w3grid != 2 && t.w3grid != 4 && t.w3grid != 8) t.w3grid = 0;
    t.big = tune_env_int("ICKA_TUNE_GEMM_BIG_TILES", 2, 0, 2);
    return t;
}();
// icka_gemm_desc.tune (include/icka_hip.h, ICKA_TUNE_*): 4 bits per field, 0 = keep the default, else the field's code.
// Returns false for a code outside a field's range (-> ICKA_E_ARG).
static bool tune_of(uint64_t bits, Tune& t) {
    t = kTuneEnv;
    if (!bits) return true;
    const int ring = (int)(bits & 15), tile = (int)((bits >> 4) & 15), wide = (int)((bits >> 8) & 15), dir = (int)((bits >> 12) & 15),
              ws = (int)((bits >> 16) & 15), grid = (int)((bits >> 20) & 15), big = (int)((bits >> 24) & 15), abl = (int)((bits >> 28) & 15);
    if (bits >> 32) return false;
    if (ring) { if (ring < 2 || ring > 5) return false; t.nbuf = ring; }
    if (tile) { if (tile > 2) return false; t.bn = tile == 1 ? 96 : 128; }
    if (wide) { if (wide > 2) return false; t.w3 = wide - 1; }
    if (dir) { if (dir > 2) return false; t.direct = dir - 1; }
    if (ws) { if (ws > 4) return false; t.ws = ws - 1; }
    if (grid) { if (grid != 1 && grid != 2 && grid != 4 && grid != 8) return false; t.w3grid = grid; }
    if (big) { if (big > 3) return false; t.big = big - 1; }
    if (abl) { if (abl > 3) return false; t.abl = abl; }
    return true;
}
#ifdef ICKA_GEMM_STAMP
static unsigned long long* g_stamp = nullptr;   // diagnostic build only: per-segment cycle sums (icka_diag_gemm_stamp_buffer)
#endif

__global__ void scale_c_kernel(float* C, int64_t ldc, int M, int N, float beta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int64_t r = i / N;
    float* p = C + r * ldc + (i - r * N);
    *p = beta == 0.f ? 0.f : *p * beta;
}


// =====================================================================================================================
// 256 x 192 output tiles, 12 waves (8 compute waves of 64 x 96 + 4 loader waves), for the wide-output / short-K GEMMs of
// the step (qkv: 4096 x 2304, ffn-up and d(ffn-down): 4096 x 3072, K = 768): 192 / 256 tiles = ONE round of the 256 CUs
// instead of 2.25 / 3 rounds of 128 x 128 tiles, and 0.75x the operand bytes per FLOP through L2 -> LDS (the in-step
// limiter).  Two compute waves per SIMD cover each other's LDS latency (fragments are read per
// 32-deep step right before their MFMAs, as in the two-blocks-per-CU kernel); LDS ring of 2 stages of
// [A rows 0..127 | A rows 128..255 | B 0..95 | B 96..191] (4 x 16 KiB images in the usual swizzled layouts); outputs go
// straight from the accumulators through the general 4-column epilogue.  A is k-contiguous (NT and NN).
constexpr int W3_A = 2 * TILE_BYTES, W3_B = 2 * TILE_BYTES;   // one k-tile of A (2 x 128 rows) / of B (2 x 96 columns)
constexpr int W3_NA = 3, W3_NB = 2;                           // ring depths: 3 x 32 KiB + 2 x 32 KiB = 160 KiB = the whole LDS
// tile of block ``bid`` of the 12-wave kernel.  Blocks b and b+8 share an XCD (and its private 4 MiB L2), and the nb / 8
// tiles an XCD works on at the same time decide what it has to fetch: a pm x pn cut of the tile grid over the XCDs makes the
// chip fetch pn * |A| + pm * |B| (every XCD needs the A row panels and the B column panels of its patch).  The host picks the
// cut (gemm_w3_grid: the dividing one with the smallest pn * M + pm * N; 8 x 1 = the row-major runs of rounds 1-2) and passes
// sn = column tiles per patch and log2(pn): ffn-up 4 x 2 (31.5 MB instead of 44 MB at c2), the M = 8192 x N = 1024 shapes of
// c4 stay 8 x 1.
__device__ __forceinline__ void w3_origin(int bid, int sn, int pnlog, int bnw, int& m0, int& n0) {
    const int xcd = bid & 7, li = bid >> 3;
    const int xi = xcd >> pnlog, xj = xcd & ((1 << pnlog) - 1);
    const int r = li / sn, c = li - r * sn;
    // sm = rows of tiles per patch = (nb / 8) / sn, implied: patch xi starts at row xi * sm
    m0 = (xi * ((int)(gridDim.x >> 3) / sn) + r) * 256;
    n0 = (xj * sn + c) * bnw;
}
template <bool B_KM, bool F16 = false, int BNW = 192>
__global__ __launch_bounds__(768) void gemm_w3_kernel(const GemmArgs gp) {
    const GemmArgs g = gp;
    __shared__ __attribute__((aligned(16))) char smem[W3_NA * W3_A + W3_NB * W3_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(char, smem)));
    // BNW = tile width: 192 (two 96-column halves) or 128 (two 64-column halves: N = 1024 at bert-large, 256 tiles = one
    // round where 128x128 tiles take two); the B images keep their 16 KiB slots either way
    constexpr int BH = BNW / 2, NB16 = BH / 16;
    static_assert(BNW == 192 || BNW == 128, "tile width");
    int m0, n0;
    w3_origin(blockIdx.x, g.w3_sn, g.w3_pnlog, BNW, m0, n0);
    const int nk = g.K / BK;
    if (wave >= 8) {
        // ------------------------------------------------------------------------------------------- loader waves
        // The activation operand A (cold, the expensive one: tools/gemm_cold.py) runs TWO k-tiles ahead in a ring of 3,
        // the weight operand B one k-tile ahead in a ring of 2.  Issue order per iteration: B(kt+1), then A(kt+2); LDS-DMA
        // returns in order, so "A(kt), B(kt) landed" = at most the 8 instructions of A(kt+1) still in flight.
        const int lw = wave - 8;
        const bf16_t* pa0[4];
        const bf16_t* pa1[4];
        const bf16_t* pb0[4];
        const bf16_t* pb1[4];
        dma_init<false>(pa0, g.A, g.lda, m0, lw, lane);
        dma_init<false>(pa1, g.A, g.lda, m0 + 128, lw, lane);
        dma_init<B_KM, BH>(pb0, g.B, g.ldb, n0, lw, lane);
        dma_init<B_KM, BH>(pb1, g.B, g.ldb, n0 + BH, lw, lane);
        const int64_t sa = BK, sb = B_KM ? (int64_t)BK * g.ldb : BK;
        constexpr int NJB = B_KM ? 4 : BH / 32;
        const uint32_t ldsB = lds0 + W3_NA * W3_A;
#define ICKA_W3_A(SLOT)                                                          \
    do {                                                                         \
        dma_issue(pa0, sa, lds0 + (SLOT) * W3_A + lw * 1024);                    \
        dma_issue(pa1, sa, lds0 + (SLOT) * W3_A + TILE_BYTES + lw * 1024);       \
    } while (0)
#define ICKA_W3_B(SLOT)                                                          \
    do {                                                                         \
        dma_issue<NJB>(pb0, sb, ldsB + (SLOT) * W3_B + lw * 1024);               \
        dma_issue<NJB>(pb1, sb, ldsB + (SLOT) * W3_B + TILE_BYTES + lw * 1024);  \
    } while (0)
        ICKA_W3_A(0);
        ICKA_W3_B(0);
        if (nk > 1) ICKA_W3_A(1);
        int sa3 = 2;   // A slot of tile kt+2
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) wait_vmcnt<8>(); else wait_vmcnt<0>();   // A(kt+1) may still be in flight (8 pieces: 2 x 4)
            __builtin_amdgcn_s_barrier();        // tile kt published; every compute wave is done with tile kt-1
            if (kt + 1 < nk) ICKA_W3_B((kt + 1) & 1);
            if (kt + 2 < nk) ICKA_W3_A(sa3);
            sa3 = sa3 == 2 ? 0 : sa3 + 1;
        }
#undef ICKA_W3_A
#undef ICKA_W3_B
    }
    // ---------------------------------------------------------------------------------------------- compute waves
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * BH;
    f32x4 acc[4][NB16];
    if (wave < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* sA = smem + (kt % W3_NA) * W3_A + (wr >> 7) * TILE_BYTES;
            const char* sB = smem + W3_NA * W3_A + (kt & 1) * W3_B + (wave & 1) * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[NB16];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = read_frag<false>(sA, (wr & 127) + 16 * t, ks, lane);
#pragma unroll
                for (int t = 0; t < NB16; ++t) fb[t] = read_frag<B_KM>(sB, 16 * t, ks, lane);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NB16; ++ni) acc[mi][ni] = mfma16t<F16>(fb[ni], fa[mi], acc[mi][ni]);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // MFMA -> VALU read wait states (see gemm_ws_body)
    }
    // ---- epilogue through LDS in two passes of 128 x 192 (96 KiB f32): in pass p EVERY compute wave deposits the 32-row
    //      half p of its 64 x 96 accumulator tile (so only 48 accumulator registers stay alive under the row loop: with
    //      whole wave tiles per pass the other pass's 96 spilled to scratch at the 168-register cap of 3 waves / SIMD),
    //      then all 12 waves finish 8-column groups of rows with 16-byte row-contiguous accesses
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();   // operand ring dead (pass 0) / previous C tile consumed (pass 1)
        if (wave < 8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = (wave >> 1) * 32 + 16 * h + (lane & 15);
#pragma unroll
                for (int ni = 0; ni < NB16; ++ni) {
                    const int ch = (wc >> 2) + 4 * ni + (lane >> 4);
                    *reinterpret_cast<f32x4*>(smem + (BNW == 192 ? off_cw(row, ch) : off_c(row, ch))) = acc[2 * pass + h][ni] * g.alpha;
                }
            }
        }
        __syncthreads();
        epilogue_rows<32, BNW / 8, BNW == 192 ? 1 : 2>(g, smem, m0 + 32 * pass, n0, tid);
    }
}

// =====================================================================================================================
// QKV projection + self-attention of a head in ONE launch (sequence length 128, head size 64 -- the reference's
// max_seq_length 128 text: BertSelfAttention.forward, Cross_Modal_Interaction_Module.py:478-506).
// The 12-wave kernel's 256 x 192 tile is laid over the problem so that its 192 columns are the 64 query, 64 key and 64 value
// columns of ONE head (B rows head*64 .. +63 of each of the three stacked weight blocks: only the loader's row pointers change)
// and its 256 rows are TWO samples: after the k-loop the block holds everything the attention of those two (sample, head) pairs
// needs.  Epilogue: accumulators + bias -> bf16 -> six [128][64] off_t images in the dead operand ring (q, k, v of either sample);
// then the four loader waves stream the images to the qkv activation (the attention backward and the weight gradient read it
// later) while the eight compute waves run the whole-head attention forward of attn_core.h straight from the images -- 32 queries
// per wave -- and store context rows and log-sum-exp.  Same bf16 q / k / v, same device function: bit for bit the result of
// icka_gemm + icka_attn_fwd, without the attention launch, its 19 MB re-read of qkv and the kernel boundary between them.
struct QkvAttnArgs { GemmArgs g; AttnArgs a; };

template <int HALF>
__device__ __forceinline__ void dma_init_qkv(const bf16_t* (&ptr)[4], const bf16_t* __restrict__ P, int64_t ld, int head, int H,
                                             int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = wave + 4 * j;                 // (piece 3 of a 96-row image is never issued)
        const int row = (8 * p + (lane >> 3)) % 96;
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const int r = 96 * HALF + row;              // tile column 0..191 = [q | k | v] of the head
        ptr[j] = P + (int64_t)((r >> 6) * H + head * 64 + (r & 63)) * ld + 8 * lc;
    }
}

// SEQ = tokens per sample: 128 (the tile's 256 rows = two samples, 32 queries per compute wave in one pass) or 256 (one sample,
// bert-large at BASELINE config c4: 32 queries per wave as two passes of 16 against the 256 keys).  F16: fp16 operands of the
// projection (the "mixed16" forward GEMMs; q / k / v stay bf16).  KB: the attention leaves its keep bits (SEQ = 256 only).
template <bool DROP, bool F16, int SEQ, bool KB>
__global__ __launch_bounds__(768) void gemm_qkv_attn_kernel(const QkvAttnArgs p) {
    static_assert(SEQ == 128 || SEQ == 256, "tokens per sample");
    const GemmArgs g = p.g;
    __shared__ __attribute__((aligned(16))) char smem[W3_NA * W3_A + W3_NB * W3_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(char, smem)));
    constexpr int BH = 96, NB16 = 6;
    int m0, n0;
    w3_origin(blockIdx.x, g.w3_sn, g.w3_pnlog, 192, m0, n0);
    const int head = n0 / 192, H = g.N / 3;
    const int nk = g.K / BK;
    if (wave >= 8) {
        // loader waves: gemm_w3_kernel's ring (A two k-tiles ahead in a ring of 3, B one ahead in a ring of 2)
        const int lw = wave - 8;
        const bf16_t* pa0[4];
        const bf16_t* pa1[4];
        const bf16_t* pb0[4];
        const bf16_t* pb1[4];
        dma_init<false>(pa0, g.A, g.lda, m0, lw, lane);
        dma_init<false>(pa1, g.A, g.lda, m0 + 128, lw, lane);
        dma_init_qkv<0>(pb0, g.B, g.ldb, head, H, lw, lane);
        dma_init_qkv<1>(pb1, g.B, g.ldb, head, H, lw, lane);
        const int64_t sa = BK, sb = BK;
        const uint32_t ldsB = lds0 + W3_NA * W3_A;
#define ICKA_QA_A(SLOT)                                                          \
    do {                                                                         \
        dma_issue(pa0, sa, lds0 + (SLOT) * W3_A + lw * 1024);                    \
        dma_issue(pa1, sa, lds0 + (SLOT) * W3_A + TILE_BYTES + lw * 1024);       \
    } while (0)
#define ICKA_QA_B(SLOT)                                                          \
    do {                                                                         \
        dma_issue<3>(pb0, sb, ldsB + (SLOT) * W3_B + lw * 1024);                 \
        dma_issue<3>(pb1, sb, ldsB + (SLOT) * W3_B + TILE_BYTES + lw * 1024);    \
    } while (0)
        ICKA_QA_A(0);
        ICKA_QA_B(0);
        if (nk > 1) ICKA_QA_A(1);
        int sa3 = 2;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) wait_vmcnt<8>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + 1 < nk) ICKA_QA_B((kt + 1) & 1);
            if (kt + 2 < nk) ICKA_QA_A(sa3);
            sa3 = sa3 == 2 ? 0 : sa3 + 1;
        }
#undef ICKA_QA_A
#undef ICKA_QA_B
    }
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * BH;
    f32x4 acc[4][NB16];
    if (wave < 8) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB16; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* sA = smem + (kt % W3_NA) * W3_A + (wr >> 7) * TILE_BYTES;
            const char* sB = smem + W3_NA * W3_A + (kt & 1) * W3_B + (wave & 1) * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[4], fb[NB16];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = read_frag<false>(sA, (wr & 127) + 16 * t, ks, lane);
#pragma unroll
                for (int t = 0; t < NB16; ++t) fb[t] = read_frag<false>(sB, 16 * t, ks, lane);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NB16; ++ni) acc[mi][ni] = mfma16t<F16>(fb[ni], fa[mi], acc[mi][ni]);
            }
        }
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // MFMA -> VALU read wait states
    }
    __syncthreads();   // the operand ring is dead: it becomes the bf16 images [SEQ][64] of q, k, v (of either sample): smem + (3 s + mat) * IMG
    constexpr int IMG = SEQ * 128;
    if (wave < 8) {
        const int s = SEQ == 128 ? wave >> 2 : 0;
#pragma unroll
        for (int ni = 0; ni < NB16; ++ni) {
            const int c0 = wc + 16 * ni + 4 * (lane >> 4);   // tile column of this lane's four accumulator columns
            const int mat = c0 >> 6, cc = c0 & 63;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(g.bias + mat * H + head * 64 + cc);
            char* im = smem + (3 * s + mat) * IMG;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int r = (wr & (SEQ - 1)) + 16 * mi + (lane & 15);
                const f32x4 v = acc[mi][ni] + bv;
                *reinterpret_cast<u32x2*>(im + off_t(r, cc >> 3) + 8 * ((cc >> 2) & 1)) = pack4(v[0], v[1], v[2], v[3]);
            }
        }
    }
    __syncthreads();
    if (wave >= 8) {
        // the qkv activation for the backward: 256 rows x 3 matrices x 8 chunks of 16 B, 24 per loader thread; a row's 64 columns
        // (128 B) are contiguous in global memory
        bf16_t* C = reinterpret_cast<bf16_t*>(g.C);
        const int t = tid - 512;
#pragma unroll 4
        for (int i = 0; i < 24; ++i) {
            const int id = t + 256 * i;
            const int im = id / (SEQ * 8), r = (id >> 3) & (SEQ - 1), c = id & 7;
            const int s = im >= 3 ? 1 : 0, mat = im - 3 * s;
            const u32x4 v = *reinterpret_cast<const u32x4*>(smem + im * IMG + off_t(r, c));
            st_main(reinterpret_cast<u32x4*>(C + (int64_t)(m0 + SEQ * s + r) * g.ldc + mat * H + head * 64 + 8 * c), v, g.plain);
        }
        return;
    }
    AttnArgs a = p.a;
    a.drop = drop_resolve(a.drop);
    a.Sq = a.Skv = SEQ;   // (the host checked it: as constants they fold the core's end-of-sequence clamps and selects away)
    const int s = SEQ == 128 ? wave >> 2 : 0, b = m0 / SEQ + s;
    const char* sQ = smem + 3 * s * IMG;
    if constexpr (SEQ == 128) {
        attn_fwd_whole_head<2, 8, DROP, false, KB>(a, sQ, sQ + IMG, sQ + 2 * IMG, 2 * (wave & 3), 0, b * a.h + head, b, head, lane);
    } else {
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            asm volatile("" ::: "memory");   // (keeps the pass's LDS reads -- K fragments, mask -- inside the pass: hoisted they spill)
            attn_fwd_whole_head<1, 16, DROP, false, KB>(a, sQ, sQ + IMG, sQ + 2 * IMG, 2 * wave + t, 0, b * a.h + head, b, head, lane);
        }
    }
}

// rows pm of the pm x pn XCD cut of a 256 x bnw tile grid that fetches least: min pn * M + pm * N over the cuts that divide it
// (forced: Tune::w3grid, where it divides the tile grid)
static int gemm_w3_grid(int M, int N, int bnw, int forced) {
    const int nbm = M / 256, nbn = N / bnw;
    int best = 8;
    long cost = -1;
    for (int pm = 8; pm >= 1; pm >>= 1) {
        const int pn = 8 / pm;
        if (nbm % pm || nbn % pn) continue;
        if (forced && pm != forced) continue;
        const long c = (long)pn * M + (long)pm * N;
        if (cost < 0 || c < cost) { cost = c; best = pm; }
    }
    if (cost < 0) {   // forced cut does not divide: fall back to the free choice
        for (int pm = 8; pm >= 1; pm >>= 1) {
            const int pn = 8 / pm;
            if (nbm % pm || nbn % pn) continue;
            const long c = (long)pn * M + (long)pm * N;
            if (cost < 0 || c < cost) { cost = c; best = pm; }
        }
    }
    return best;
}

template <bool A_KM, bool B_KM, bool F16 = false>
int launch(GemmArgs g, bool aligned, hipStream_t st, const Tune& t) {
    const int nb = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    if constexpr (!A_KM && !B_KM) {
        if (!aligned && g.n64ok && t.ws) {   // 64-channel convolutions of the ResNet stem / layer1: 128x64 tiles
            hipLaunchKernelGGL((gemm_ws_kernel<false, false, 3, 0, 64, F16>), dim3((g.M / BM) * (g.N / 64)), dim3(512), 0, st, g);
            ICKA_CHECK_LAUNCH();
            return 0;
        }
    }
    if (aligned) {
#ifdef ICKA_GEMM_ABLATE
        if (t.abl == 1 && !t.ws) hipLaunchKernelGGL((gemm_dma_kernel<A_KM, B_KM, 3, 1>), dim3(nb), dim3(256), 0, st, g);
        else if (t.abl == 2 && !t.ws) hipLaunchKernelGGL((gemm_dma_kernel<A_KM, B_KM, 3, 2>), dim3(nb), dim3(256), 0, st, g);
        else
#endif
        {
            // ring depth: many tiles per CU -> two co-resident blocks (64 KiB ring of 2) overlap one block's
            // epilogue with the other's main loop; few tiles -> one block per CU with a deeper ring (measured,
            // tools/gemm_bench.py)
            if (t.ws || F16) {   // (fp16 operands exist on the warp-specialised kernels only)
#ifdef ICKA_GEMM_ABLATE
                if (t.bn == 96 && g.n96ok) {   // the 128x96-tile kernel (ICKA_TUNE_TILE_N(96))
                    const int nb96 = (g.M / BM) * (g.N / 96);
                    if (t.abl == 1) { hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 1, 96>), dim3(nb96), dim3(512), 0, st, g); ICKA_CHECK_LAUNCH(); return 0; }
                    if (t.abl == 2) { hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 2, 96>), dim3(nb96), dim3(512), 0, st, g); ICKA_CHECK_LAUNCH(); return 0; }
                    if (t.abl == 3) { hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 3, 96>), dim3(nb96), dim3(512), 0, st, g); ICKA_CHECK_LAUNCH(); return 0; }
                }
                if (t.abl == 1) { hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 1>), dim3(nb), dim3(512), 0, st, g); ICKA_CHECK_LAUNCH(); return 0; }
                if (t.abl == 2) { hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 2>), dim3(nb), dim3(512), 0, st, g); ICKA_CHECK_LAUNCH(); return 0; }
#endif
                // 256x192 tiles (12-wave kernel) where they cover the CUs in ONE round: wide outputs with a short reduction
                if constexpr (!A_KM) {
                    const int nb3 = (g.M / 256) * (g.N / 192);
                    // ... or in whole rounds: at least 3/4 of the last round of 256 must be filled
                    const int rounds3 = (nb3 + 255) / 256;
                    if (t.w3 && g.M % 256 == 0 && g.N % 192 == 0 && g.K <= 1024 && g.K1 == 0 && nb3 % 8 == 0 &&
                        nb3 >= 128 && 4 * nb3 >= 3 * 256 * rounds3 && g.ksplit == 1) {
                        { const int pm = gemm_w3_grid(g.M, g.N, 192, t.w3grid), pn = 8 / pm;
                          g.w3_sn = (g.N / 192) / pn; g.w3_pnlog = pn == 1 ? 0 : (pn == 2 ? 1 : (pn == 4 ? 2 : 3)); }
                        hipLaunchKernelGGL((gemm_w3_kernel<B_KM, F16>), dim3(nb3), dim3(768), 0, st, g);
                        ICKA_CHECK_LAUNCH();
                        return 0;
                    }
                    // 256x128 tiles (the same kernel, 64-column halves) where 128x128 tiles would take exactly two rounds of
                    // the CUs with one block each (N = 1024 at bert-large / M = 8192: 256 tiles), any K
                    const int nb2 = (g.M / 256) * (g.N / 128);
                    if (t.w3 && g.M % 256 == 0 && g.K1 == 0 && g.ksplit == 1 && nb2 % 8 == 0 && nb2 >= 192 && nb2 <= 256 &&
                        !(g.n96ok && t.bn == 96) && !(g.K <= 1024 && g.direct && g.c_f32 && g.epi == ICKA_EPI_NONE && g.beta == 0.f)) {
                        // (short reductions with a plain f32 output stay on the 128-wide kernel: its direct epilogue beats
                        //  the two staged passes here, 25.9 vs 28.3 us at 8192 x 1024 x 1024)
                        { const int pm = gemm_w3_grid(g.M, g.N, 128, t.w3grid), pn = 8 / pm;
                          g.w3_sn = (g.N / 128) / pn; g.w3_pnlog = pn == 1 ? 0 : (pn == 2 ? 1 : (pn == 4 ? 2 : 3)); }
                        hipLaunchKernelGGL((gemm_w3_kernel<B_KM, F16, 128>), dim3(nb2), dim3(768), 0, st, g);
                        ICKA_CHECK_LAUNCH();
                        return 0;
                    }
                }
                // Tile width: 128x96 tiles when they quantise better onto the 256 CUs (N = 768: 256 tiles instead of 192).
                if (g.n96ok && t.bn != 128) {
                    // measured (profiles/README.md): a 128x96 tile costs ~0.9-1.0 of a 128x128 one (the k-loop is bound
                    // by LDS / L1 traffic of the A tile, not by MFMA count), so the narrow tile only pays where it turns
                    // an under-filled single round into a full one
                    const int nb96 = (g.M / BM) * (g.N / 96);
                    if (t.bn == 96 || (nb < 256 && nb96 <= 256 && nb96 > nb)) {
                        if ((t.ws == 2 || (t.ws == 1 && nb96 >= 448 && g.K <= 1024)) && !A_KM)
                            hipLaunchKernelGGL((gemm_ws2_kernel<A_KM, B_KM, 96, F16>), dim3(nb96), dim3(512), 0, st, g);
                        else if (t.nbuf == 4 && !F16)
                            hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 4, 0, 96>), dim3(nb96), dim3(512), 0, st, g);
                        else if (t.nbuf == 5 && !F16)
                            hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 5, 0, 96>), dim3(nb96), dim3(512), 0, st, g);
                        else
                            hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 0, 96, F16>), dim3(nb96), dim3(512), 0, st, g);
                        ICKA_CHECK_LAUNCH();
                        return 0;
                    }
                }
                // measured (tools/gemm_bench.py): two co-resident blocks win only for short reductions on grids of
                // >= ~2 tiles per CU (qkv, ffn-up, d-ffn-down); long-K shapes prefer the deeper ring of one block
                if ((t.ws == 2 || (t.ws == 1 && nb >= 448 && g.K <= 1024)) && !A_KM)
                    hipLaunchKernelGGL((gemm_ws2_kernel<A_KM, B_KM, 128, F16>), dim3(nb), dim3(512), 0, st, g);
                else if (t.nbuf == 4 && !F16) hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 4>), dim3(nb), dim3(512), 0, st, g);
                else if (t.nbuf == 5 && !F16) hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 5>), dim3(nb), dim3(512), 0, st, g);
                else hipLaunchKernelGGL((gemm_ws_kernel<A_KM, B_KM, 3, 0, 128, F16>), dim3(nb), dim3(512), 0, st, g);
                ICKA_CHECK_LAUNCH();
                return 0;
            }
            const int nbuf = t.nbuf > 0 ? t.nbuf : (nb >= 448 ? 2 : 4);
            if (nbuf == 2) hipLaunchKernelGGL((gemm_dma_kernel<A_KM, B_KM, 2, 0>), dim3(nb), dim3(256), 0, st, g);
            else if (nbuf == 3) hipLaunchKernelGGL((gemm_dma_kernel<A_KM, B_KM, 3, 0>), dim3(nb), dim3(256), 0, st, g);
            else hipLaunchKernelGGL((gemm_dma_kernel<A_KM, B_KM, 4, 0>), dim3(nb), dim3(256), 0, st, g);
        }
    } else {
        // skinny output + long reduction (classifier weight gradient: 13 x 768 x 4096 tokens): split K over
        // blockIdx.y and accumulate f32 partial tiles atomically (C is scaled by beta / zeroed first)
        const int nk = (g.K + BK - 1) / BK;
        int ks = 1;
        // (M <= 256 only: weight-gradient shapes.  Forward outputs such as the [tokens, labels] emissions stay on one
        //  block per tile so that two identical calls give bitwise identical logits -- atomic split-K order flipped
        //  Viterbi near-ties between a 'dev' and a 'test' pass of the same batch.)
        if (g.c_f32 && g.epi == ICKA_EPI_NONE && nb < 64 && nk >= 16 && g.M <= 256 && !g.C3) {   // (a wire copy needs the final value in one epilogue)
            ks = 256 / nb;
            if (ks > nk / 4) ks = nk / 4;
            if (ks < 1) ks = 1;
        }
        if (ks > 1) {
            const int64_t n = (int64_t)g.M * g.N;
            hipLaunchKernelGGL(scale_c_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               reinterpret_cast<float*>(g.C), g.ldc, g.M, g.N, g.beta);
            ICKA_CHECK_LAUNCH();
            g.beta = 0.f;
            g.ksplit = ks;
        }
        hipLaunchKernelGGL((gemm_kernel<A_KM, B_KM, false, F16>), dim3(nb, ks), dim3(256), 0, st, g);
    }
    ICKA_CHECK_LAUNCH();
    return 0;
}

inline bool vec_ok(const void* p, int64_t ld) { return (ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(p) & 15) == 0); }

}  // namespace

#ifdef ICKA_GEMM_STAMP
// diagnostic builds only (tools/gemm_stamp.py): device buffer of [blocks][8] u64 receiving per-segment cycle sums
extern "C" int icka_diag_gemm_stamp_buffer(void* p) {
    g_stamp = (unsigned long long*)p;
    return 0;
}
#endif

// Diagnostic site classes of the per-site store-policy matrix (profiles/r04_store_policy_matrix.txt): which producer of a
// BERT layer a launch is, from its op / epilogue / shape ratio (N, K in units of H = min(N, K) work for bert-base and -large).
//   bit 0 QKV (NT, N = 3H)            -> attention          bit 4 d(ffn-down) dgrad (NN, N = 4H, GELU' epilogue) -> d(ffn-up), wgrad
//   bit 1 out-proj (NT, N = K)        -> LayerNorm          bit 5 d(ffn-up) dgrad   (NN, K = 4N)               -> LayerNorm bwd
//   bit 2 ffn-up + GELU (NT, N = 4K)  -> ffn-down           bit 6 d(out-proj) dgrad (NN, N = K)                -> attention bwd
//   bit 3 ffn-down (NT, K = 4N)       -> LayerNorm          bit 7 d(QKV) dgrad      (NN, K = 3N)               -> LayerNorm bwd
static int site_bit(const icka_gemm_desc* d) {
    const int64_t N = d->N, K = d->K;
    if (d->c_is_f32 == 1 || d->M < 1024) return -1;
    if (d->op == ICKA_GEMM_NT) {
        if (N == 3 * K) return 0;
        if (N == K) return 1;
        if (N == 4 * K) return 2;
        if (K == 4 * N) return 3;
    } else if (d->op == ICKA_GEMM_NN) {
        if (N == 4 * K) return 4;
        if (K == 4 * N) return 5;
        if (N == K) return 6;
        if (K == 3 * N) return 7;
    }
    return -1;
}
static int plain_mask() {
    static int m = -1;
    if (m < 0) {
        const char* e = getenv("ICKA_GEMM_PLAIN_MASK");
        m = e ? (int)strtol(e, nullptr, 0) : 0;
    }
    return m;
}

static int convert(const icka_gemm_desc* d, GemmArgs& g, bool& aligned, Tune& t) {
    if (!d || !d->A || !d->B || !d->C) return ICKA_E_ARG;
    if (!tune_of(d->tune, t)) return ICKA_E_ARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return ICKA_E_SHAPE;
    if (d->op < ICKA_GEMM_NT || d->op > ICKA_GEMM_TN) return ICKA_E_ARG;
    if (d->K1 != 0 && (d->K1 < 0 || d->K1 >= d->K || d->K1 % BK != 0 || !d->A2 || !d->B2)) return ICKA_E_ARG;
    if ((d->epilogue == ICKA_EPI_GELU) && !d->C2) return ICKA_E_ARG;
    if ((d->epilogue == ICKA_EPI_DGELU || d->epilogue == ICKA_EPI_ADD || d->epilogue == ICKA_EPI_GATE ||
         d->epilogue == ICKA_EPI_ADD_RELU) && !d->aux)
        return ICKA_E_ARG;
    g.M = d->M; g.N = d->N; g.K = d->K; g.K1 = d->K1;
    g.A = (const bf16_t*)d->A; g.lda = d->lda; g.B = (const bf16_t*)d->B; g.ldb = d->ldb;
    g.A2 = (const bf16_t*)d->A2; g.lda2 = d->lda2; g.B2 = (const bf16_t*)d->B2; g.ldb2 = d->ldb2;
    g.C = d->C; g.ldc = d->ldc; g.C2 = (bf16_t*)d->C2; g.ldc2 = d->ldc2;
    g.aux = (const bf16_t*)d->aux; g.ldaux = d->ldaux; g.bias = d->bias; g.bias2 = d->bias2;
    g.alpha = d->alpha; g.beta = d->beta; g.epi = d->epilogue;
    if (d->c_is_f32 < 0 || d->c_is_f32 > 2) return ICKA_E_ARG;
    g.c_f32 = d->c_is_f32 == 1; g.c_f16 = d->c_is_f32 == 2;
    g.f16 = d->ab_f16 != 0; g.C3 = (bf16_t*)d->C3; g.ldc3 = d->ldc3; g.aux_f16 = d->aux_f16 != 0;
    if (g.f16 && d->op != ICKA_GEMM_NT) return ICKA_E_ARG;       // fp16 operands: forward (NT) GEMMs only
    if (g.c_f16 && d->beta != 0.f) return ICKA_E_ARG;              // fp16 outputs are never accumulated into
    if (g.C3 && !g.c_f16 && !g.c_f32) return ICKA_E_ARG;           // C3 = bf16 twin of an fp16 or an f32 main output
    // the wire copy of an f32 output is compiled into the weight-gradient (TN) instances of the fast paths only: any other
    // op would return 0 and leave the wire buffer stale
    if (g.C3 && g.c_f32 && d->op != ICKA_GEMM_TN) return ICKA_E_ARG;
    g.c3_only = d->c3_only != 0;
    {
        const int sb = site_bit(d);
        g.plain = (sb >= 0 && ((plain_mask() >> sb) & 1)) ? 1 : 0;
    }
    if (g.c3_only && !(g.C3 && g.c_f32 && d->beta == 0.f && d->epilogue == ICKA_EPI_NONE)) return ICKA_E_ARG;
    if (g.c_f16 && d->colsum_out) return ICKA_E_ARG;
    g.abl = t.abl;
    g.stamp = nullptr;
#ifdef ICKA_GEMM_STAMP
    g.stamp = g_stamp;
    {   // diagnostic builds: ICKA_GEMM_STAMP_FILTER="op,N,K" stamps only that shape (all GEMMs share one buffer)
        static int f_op = -2, f_n = 0, f_k = 0;
        if (f_op == -2) {
            f_op = -1;
            if (const char* e = getenv("ICKA_GEMM_STAMP_FILTER")) sscanf(e, "%d,%d,%d", &f_op, &f_n, &f_k);
        }
        if (f_op >= 0 && !(d->op == f_op && d->N == f_n && d->K == f_k)) g.stamp = nullptr;
    }
#endif
    g.colsum = d->colsum_out;
    g.colsum_acc = d->colsum_accumulate;
    g.ksplit = 1;
    g.n96ok = 0;
    g.n64ok = 0;
    g.direct = t.direct;
    g.a_vec = vec_ok(d->A, d->lda) && (d->K1 == 0 || vec_ok(d->A2, d->lda2));
    g.b_vec = vec_ok(d->B, d->ldb) && (d->K1 == 0 || vec_ok(d->B2, d->ldb2));
    auto al = [](const void* p, int64_t ld, int64_t mod) {
        return !p || (((reinterpret_cast<uintptr_t>(p) & 15) == 0) && ld % mod == 0);
    };
    if (d->colsum_out && d->op != ICKA_GEMM_TN) return ICKA_E_ARG;
    aligned = (d->M % BM == 0) && (d->N % BN == 0) && (d->K % BK == 0) && g.a_vec && g.b_vec &&
              al(d->C, d->ldc, d->c_is_f32 == 1 ? 4 : 8) && al(d->C2, d->ldc2, 8) && al(d->aux, d->ldaux, 8) &&
              al(d->bias, 4, 4) && al(d->bias2, 4, 4) && al(d->C3, d->ldc3, 8);
    g.n96ok = aligned && d->N % 96 == 0;
    g.n64ok = !aligned && (d->M % BM == 0) && (d->N % 64 == 0) && (d->K % BK == 0) && g.a_vec && g.b_vec &&
              al(d->C, d->ldc, d->c_is_f32 == 1 ? 4 : 8) && al(d->C2, d->ldc2, 8) && al(d->aux, d->ldaux, 8) &&
              al(d->bias, 4, 4) && al(d->bias2, 4, 4) && al(d->C3, d->ldc3, 8) && d->K1 == 0 && !d->colsum_out;
    return 0;
}

extern "C" int icka_gemm(const icka_gemm_desc* d, void* stream) {
    GemmArgs g;
    Tune t;
    bool aligned = false;
    const int rc = convert(d, g, aligned, t);
    if (rc) return rc;
    if (g.colsum && !(aligned && t.ws)) return ICKA_E_ARG;   // fused column sums exist on the warp-specialised path
    hipStream_t st = (hipStream_t)stream;
    if (g.f16) return launch<false, false, true>(g, aligned, st, t);
    switch (d->op) {
        case ICKA_GEMM_NT: return launch<false, false>(g, aligned, st, t);
        case ICKA_GEMM_NN: return launch<false, true>(g, aligned, st, t);
        default: return launch<true, true>(g, aligned, st, t);
    }
}

static int device_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else { n = 1; (void)hipGetLastError(); }
    }
    return n;
}
constexpr int GEMM_LN_MAX_STRIPES = 64;
static int g_gemm_ln_polls = 0;      // icka_gemm_ln_test_hooks: 0 = the default budget (~7 ms of s_sleep polls)
static int g_gemm_ln_drop = -1;      // icka_gemm_ln_test_hooks: block that never arrives (-1: off)
extern "C" int64_t icka_gemm_ln_sync_words(void) { return (int64_t)GEMM_LN_MAX_STRIPES * 32; }
extern "C" int icka_gemm_ln_test_hooks(int32_t polls, int32_t drop_block) {
    g_gemm_ln_polls = polls > 0 ? polls : 0;
    g_gemm_ln_drop = drop_block >= 0 ? drop_block : -1;
    return 0;
}
extern "C" int icka_gemm_ln(const icka_gemm_desc* d, const float* bias, const void* residual, int64_t ldr, int32_t res_kind,
                            const float* gamma, const float* beta, void* y, int64_t ldy, void* y_twin, int32_t twin_f16,
                            void* xhat, float* rstd, float eps, float p_drop, uint64_t seed, uint32_t* sync_words,
                            uint32_t* error_word, void* stream) {
    if (!d || !gamma || !beta || !y || !sync_words || !error_word) return ICKA_E_ARG;
    if (res_kind < 0 || res_kind > 2) return ICKA_E_ARG;
    GemmArgs g;
    Tune t;
    bool aligned = false;
    const int rc = convert(d, g, aligned, t);
    if (rc) return rc;
    // eligibility (else ICKA_E_SHAPE: the caller takes icka_gemm + icka_ln_fwd): NT, bf16 operands, plain f32 output that the
    // direct epilogue stores, one reduction segment, EIGHT column tiles per stripe, whole groups of 8 stripes, and at most one
    // block per CU -- the blocks of a stripe wait for each other, so all of them must be resident at once
    if (d->op != ICKA_GEMM_NT || g.f16 || !aligned || !g.c_f32 || g.epi != ICKA_EPI_NONE || g.beta != 0.f || g.alpha != 1.f || g.bias ||
        g.bias2 || g.K1 != 0 || g.colsum || g.C3 || !t.direct || !t.ws || d->tune)
        return ICKA_E_SHAPE;
    const int bnt = (g.N % 96 == 0 && g.N / 96 == 8) ? 96 : ((g.N % 128 == 0 && g.N / 128 == 8) ? 128 : 0);
    const int stripes = g.M / BM;
    // (any number of stripes: where M / 128 is not a multiple of 8 a stripe's blocks spread over two or three XCDs -- the hand-off
    //  is write-through stores + agent-scope counter + L1-bypassing loads, correct under any placement, a little slower there)
    if (!bnt || stripes < 1 || stripes > GEMM_LN_MAX_STRIPES || stripes * 8 > device_cus() - g_icka_reserved_cus) return ICKA_E_SHAPE;
    if (g.N % 8 || ldy % 8 || (residual && ldr % 8) || g.ldc % 4) return ICKA_E_ALIGN;
    if ((int64_t)g.M * g.ldc * 4 >= (1ll << 31) - 64) return ICKA_E_SHAPE;      // 32-bit byte offsets of the raw buffer accesses
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(y) || (residual && !al16(residual)) || (xhat && !al16(xhat)) || (bias && !al16(bias)) || !al16(gamma) || !al16(beta) ||
        (y_twin && !al16(y_twin)))
        return ICKA_E_ALIGN;
    GemmLnArgs p;
    p.g = g;
    p.ln = LnFwdArgs{g.C, g.ldc, 1, bias, residual, ldr, res_kind, gamma, beta, (bf16_t*)y, ldy, nullptr, 0, y_twin, (bf16_t*)xhat, rstd,
                     g.M, g.N, eps, make_drop(p_drop, seed), twin_f16};
    p.sync = sync_words;
    p.err = error_word;
    p.polls = g_gemm_ln_polls > 0 ? g_gemm_ln_polls : (1 << 18);
    p.test_drop = g_gemm_ln_drop;
    hipStream_t st = (hipStream_t)stream;
    if (bnt == 96) hipLaunchKernelGGL((gemm_ln_kernel<96>), dim3(stripes * 8), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((gemm_ln_kernel<128>), dim3(stripes * 8), dim3(512), 0, st, p);
    ICKA_CHECK_LAUNCH();
    return 0;
}

// QKV projection + whole-head self-attention in one launch (gemm_qkv_attn_kernel).  d: the NT projection [M, 3 H] = x . Wqkv^T
// with its bias and a bf16 output (the stacked [q | k | v] activation, written as by icka_gemm); operands bf16, or both fp16
// ("mixed16").  The attention arguments as icka_attn_fwd_ex with Q / K / V = the three column blocks of that output.
// ICKA_E_SHAPE = not a shape this launch covers (the caller runs icka_gemm + icka_attn_fwd_ex, which give the same bits):
// S = 128 or 256 tokens per sample, head size 64, whole 256-row tiles, a tile grid of the 12-wave kernel.
template <bool F16, int SEQ>
static void launch_qkv_attn(const QkvAttnArgs& p, int nb3, hipStream_t st) {
    if (p.a.drop.thr && p.a.keepbits) {
        if constexpr (SEQ == 256) hipLaunchKernelGGL((gemm_qkv_attn_kernel<true, F16, 256, true>), dim3(nb3), dim3(768), 0, st, p);
    } else if (p.a.drop.thr) hipLaunchKernelGGL((gemm_qkv_attn_kernel<true, F16, SEQ, false>), dim3(nb3), dim3(768), 0, st, p);
    else hipLaunchKernelGGL((gemm_qkv_attn_kernel<false, F16, SEQ, false>), dim3(nb3), dim3(768), 0, st, p);
}
extern "C" int icka_gemm_qkv_attn(const icka_gemm_desc* d, const float* add_mask, void* ctx, void* ctx_f16, int64_t ldo, float* lse,
                                  int32_t B, int32_t heads, int32_t S, float scale, float p_drop, uint64_t seed, void* keep_bits,
                                  void* stream) {
    if (!d || !add_mask || !ctx) return ICKA_E_ARG;
    if (B <= 0 || heads <= 0 || S <= 0) return ICKA_E_SHAPE;
    GemmArgs g;
    Tune t;
    bool aligned = false;
    const int rc = convert(d, g, aligned, t);
    if (rc) return rc;
    if (d->op != ICKA_GEMM_NT || !aligned || g.c_f32 || g.c_f16 || g.epi != ICKA_EPI_NONE || g.beta != 0.f || g.alpha != 1.f ||
        !g.bias || g.bias2 || g.K1 != 0 || g.colsum || g.C3 || g.C2 || !t.w3 || !t.ws || d->tune)
        return ICKA_E_SHAPE;
    const int H = heads * 64;
    if ((S != 128 && S != 256) || g.N != 3 * H || g.M != B * S || g.M % 256 || g.K > 1024) return ICKA_E_SHAPE;
    const int nb3 = (g.M / 256) * heads;
    if (nb3 % 8 || nb3 < 128) return ICKA_E_SHAPE;   // (small grids: the two launches, whose attention spreads over more CUs)
    if ((int64_t)B * heads * S * S >= (1ll << 32)) return ICKA_E_SHAPE;
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (!al16(ctx) || (ctx_f16 && !al16(ctx_f16)) || ldo % 8 || !al16(g.bias)) return ICKA_E_ALIGN;
    const bool keep = keep_bits && p_drop > 0.f;
    if (keep && S != 256) return ICKA_E_SHAPE;       // (keep bits are left by the 256-token instance only: ops.ATTN_KEEPBITS "auto")
    { const int pm = gemm_w3_grid(g.M, g.N, 192, t.w3grid), pn = 8 / pm;
      if ((g.M / 256) % pm || heads % pn) return ICKA_E_SHAPE;
      g.w3_sn = heads / pn; g.w3_pnlog = pn == 1 ? 0 : (pn == 2 ? 1 : (pn == 4 ? 2 : 3)); }
    QkvAttnArgs p;
    p.g = g;
    p.a = AttnArgs{};
    const bf16_t* C = reinterpret_cast<const bf16_t*>(g.C);
    p.a.Q = C; p.a.ldq = g.ldc; p.a.K = C + H; p.a.ldk = g.ldc; p.a.V = C + 2 * H; p.a.ldv = g.ldc;
    p.a.mask = add_mask; p.a.Ow = (bf16_t*)ctx; p.a.Ow16 = (_Float16*)ctx_f16; p.a.ldo = ldo; p.a.lse = lse;
    p.a.B = B; p.a.h = heads; p.a.Sq = S; p.a.Skv = S; p.a.scale = scale; p.a.drop = make_drop(p_drop, seed);
    p.a.keepbits = p.a.drop.thr ? (uint32_t*)keep_bits : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (g.f16) { if (S == 128) launch_qkv_attn<true, 128>(p, nb3, st); else launch_qkv_attn<true, 256>(p, nb3, st); }
    else { if (S == 128) launch_qkv_attn<false, 128>(p, nb3, st); else launch_qkv_attn<false, 256>(p, nb3, st); }
    ICKA_CHECK_LAUNCH();
    return 0;
}

// 3x3 / pad 1 convolution (stride 1 or 2) of an NHWC bf16 activation as an implicit GEMM on the warp-specialised kernel:
//   y[m, co] = epilogue( sum_{tap, c} x[pixel(m) + tap][c] * w[co][tap * C + c] + bias[co] (+ aux[m, co]) )
// m = output pixel (b, oy, ox) row-major, rows [rows_valid, rows_padded) of y are written from zero patches.
// Replaces icka_conv_im2col3x3 + icka_gemm: the [rows, 9 C] patch matrix (9x the activation) is never written or read.
extern "C" int icka_conv3x3_gemm(const void* x, const void* w, const float* bias, const void* aux, int64_t ldaux, void* y,
                                 int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t stride,
                                 int64_t rows_padded, int32_t epilogue, const void* zeros, void* stream) {
    if (!x || !w || !y || !zeros) return ICKA_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 64 || Cout <= 0 || Cout % 64 || (stride != 1 && stride != 2))
        return ICKA_E_SHAPE;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const int64_t rows = (int64_t)B * Ho * Wo;
    if (rows_padded < rows || rows_padded % BM || rows_padded > 0x7fffffff) return ICKA_E_SHAPE;
    if (epilogue != ICKA_EPI_NONE && epilogue != ICKA_EPI_RELU && epilogue != ICKA_EPI_ADD_RELU && epilogue != ICKA_EPI_ADD)
        return ICKA_E_ARG;
    if ((epilogue == ICKA_EPI_ADD_RELU || epilogue == ICKA_EPI_ADD) && !aux) return ICKA_E_ARG;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!al(x) || !al(w) || !al(y) || !al(zeros) || (aux && (!al(aux) || ldaux % 8)) || (bias && !al(bias))) return ICKA_E_ALIGN;
    GemmArgs g{};
    g.M = (int)rows_padded; g.N = Cout; g.K = 9 * C; g.K1 = 0;
    g.A = (const bf16_t*)x; g.lda = C; g.B = (const bf16_t*)w; g.ldb = 9 * (int64_t)C;
    g.C = y; g.ldc = Cout; g.aux = (const bf16_t*)aux; g.ldaux = ldaux; g.bias = bias;
    g.alpha = 1.f; g.beta = 0.f; g.epi = epilogue; g.c_f32 = 0; g.a_vec = g.b_vec = 1; g.ksplit = 1; g.direct = kTuneEnv.direct;
    g.cvH = H; g.cvW = W; g.cvC = C; g.cvS = stride; g.cvHo = Ho; g.cvWo = Wo; g.cvRows = (int)rows;
    g.cvZero = (const bf16_t*)zeros;
    hipStream_t st = (hipStream_t)stream;
    if (Cout % 128 == 0)
        hipLaunchKernelGGL((gemm_ws_kernel<false, false, 3, 0, 128, false, true>), dim3((g.M / BM) * (Cout / 128)), dim3(512), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_ws_kernel<false, false, 3, 0, 64, false, true>), dim3((g.M / BM) * (Cout / 64)), dim3(512), 0, st, g);
    ICKA_CHECK_LAUNCH();
    return 0;
}

template <bool A_KM, bool B_KM, int NBUF>
__global__ __launch_bounds__(512) void gemm_ws_group_kernel(const GroupArgs ga) {
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE_BYTES];
    const int bid = blockIdx.x;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) pi += bid >= ga.start[i] ? 1 : 0;
    const GemmArgs g = ga.p[pi];
    gemm_ws_body<A_KM, B_KM, NBUF>(g, smem, bid - ga.start[pi], ga.start[pi + 1] - ga.start[pi]);
}

// =====================================================================================================================
// 256x128 output tiles for the grouped weight-gradient launch (TN, K = tokens = 4096 at c2).
// The 128x128 loop is LDS-bound: four 64x64 wave tiles read 64 KB of fragments per k-tile and the LDS-DMA writes 32 KB
// (96 KB at 128 B/clk = 768 cycles against 544 cycles of MFMA).  With 128x64 wave tiles on a 256x128 block tile the
// same four compute waves issue twice the MFMAs (1024 cycles) for 96 + 48 KB of LDS traffic (1125 cycles): per FLOP
// the LDS moves 0.73x less, and the four weight-gradient problems of a BERT layer become 216 tiles = one round of
// the 256 CUs instead of 432 tiles = two.  The bias-gradient column sums do not ride on the MFMAs here (the compute
// waves have no registers to spare: 128 accumulators + two fragment sets): they are extra blocks at the END of the
// same grid, which run on the CUs the 216 tiles leave idle and stream the dY operands once.
// Outputs are written straight from the accumulators (f32, optional beta accumulate, no epilogue operand): the only
// form the weight-gradient path needs; anything else uses the 128x128 group kernel.
constexpr int BIG_STAGE = 3 * TILE_BYTES;   // A rows 0..127 | A rows 128..255 | B   (k-major images)
constexpr int BIG_NBUF = 3;
constexpr int CS_COLS = 64;                 // operand columns per column-sum block

constexpr int MAX_RED = 4;
struct SlabRed {   // out[slot][c] (+)= sum_b partials[b*slab_stride + slot*H + c]
    const float* partials; float* out[4];
    int nslab, H, nslots, slab_stride, accumulate, blocks;
};
struct BigGroupArgs {
    GemmArgs p[MAX_GROUP];
    int start[MAX_GROUP + 1];     // first GEMM tile of each problem; start[n..] = total tiles
    int cs_start[MAX_GROUP + 1];  // first column-sum block of each problem (relative to the total tiles)
    SlabRed red[MAX_RED];         // slab reductions riding on the launch (LayerNorm dgamma / dbeta of the layer)
    int red_start[MAX_RED + 1];   // first block of each reduction (relative to tiles + column-sum blocks)
};

// 64 output values per block (8 slab lanes x 64 columns), fixed summation order: bitwise reproducible
__device__ __forceinline__ void slab_reduce_block(const SlabRed& r, char* smem, int rb) {
    float* red = reinterpret_cast<float*>(smem);   // [8][64]
    const int tid = threadIdx.x, cx = tid & 63, sy = tid >> 6;
    const int idx = rb * 64 + cx;
    const bool ok = idx < r.nslots * r.H;
    float s = 0.f;
    if (ok) {
        const int slot = idx / r.H, c = idx - slot * r.H;
        const float* p = r.partials + (int64_t)slot * r.H + c;
        for (int b = sy; b < r.nslab; b += 8) s += p[(int64_t)b * r.slab_stride];
    }
    red[sy * 64 + cx] = s;
    __syncthreads();
    if (sy == 0 && ok) {
        const int slot = idx / r.H, c = idx - slot * r.H;
        float* out = r.out[slot];
        if (out) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k * 64 + cx];
            out[c] = r.accumulate ? out[c] + t : t;
        }
    }
}
__global__ __launch_bounds__(512) void slab_reduce_kernel(const SlabRed r) {
    __shared__ __attribute__((aligned(16))) char smem[8 * 64 * 4];
    slab_reduce_block(r, smem, blockIdx.x);
}
// The 128x128 group kernel with slab reductions riding along as extra blocks at the end of the grid (as on the 256x128 launch):
// a group of few tiles (the two gate weight gradients of the head: 72 tiles) leaves most CUs idle, the reductions run there.
struct GroupArgsRed { GroupArgs ga; SlabRed red[MAX_RED]; int red_start[MAX_RED + 1]; };
template <bool A_KM, bool B_KM, int NBUF>
__global__ __launch_bounds__(512) void gemm_ws_group_red_kernel(const GroupArgsRed gr) {
    __shared__ __attribute__((aligned(16))) char smem[NBUF * 2 * TILE_BYTES];
    const int bid = blockIdx.x;
    const int tiles = gr.ga.start[MAX_GROUP];
    if (bid >= tiles) {
        const int rb = bid - tiles;
        int ri = 0;
#pragma unroll
        for (int i = 1; i < MAX_RED; ++i) ri += rb >= gr.red_start[i] ? 1 : 0;
        slab_reduce_block(gr.red[ri], smem, rb - gr.red_start[ri]);
        return;
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) pi += bid >= gr.ga.start[i] ? 1 : 0;
    const GemmArgs g = gr.ga.p[pi];
    gemm_ws_body<A_KM, B_KM, NBUF>(g, smem, bid - gr.ga.start[pi], gr.ga.start[pi + 1] - gr.ga.start[pi]);
}
// several reductions in ONE launch (the slabs no weight-gradient launch took along: each used to pay its own launch)
struct SlabRedMulti { SlabRed red[MAX_RED]; int red_start[MAX_RED + 1]; };
__global__ __launch_bounds__(512) void slab_reduce_multi_kernel(const SlabRedMulti m) {
    __shared__ __attribute__((aligned(16))) char smem[8 * 64 * 4];
    const int rb = blockIdx.x;
    int ri = 0;
#pragma unroll
    for (int i = 1; i < MAX_RED; ++i) ri += rb >= m.red_start[i] ? 1 : 0;
    slab_reduce_block(m.red[ri], smem, rb - m.red_start[ri]);
}

template <bool WIRE>
__device__ __forceinline__ void gemm_big_tn_body(const GemmArgs& g, char* smem, const int m0, const int n0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(char, smem)));
    const int nk = g.K / BK;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------- loader waves
        const int lw = wave - 4;
        const bf16_t* pa0[4];
        const bf16_t* pa1[4];
        const bf16_t* pb[4];
        dma_init<true>(pa0, g.A, g.lda, m0, lw, lane);
        dma_init<true>(pa1, g.A, g.lda, m0 + 128, lw, lane);
        dma_init<true>(pb, g.B, g.ldb, n0, lw, lane);
        const int64_t sa = (int64_t)BK * g.lda, sb = (int64_t)BK * g.ldb;
#define ICKA_BIG_STAGE(BUF)                                              \
    do {                                                                 \
        dma_issue(pa0, sa, lds0 + (BUF) + lw * 1024);                    \
        dma_issue(pa1, sa, lds0 + (BUF) + TILE_BYTES + lw * 1024);       \
        dma_issue(pb, sb, lds0 + (BUF) + 2 * TILE_BYTES + lw * 1024);    \
    } while (0)
#pragma unroll
        for (int t = 0; t < BIG_NBUF - 1; ++t)
            if (t < nk) ICKA_BIG_STAGE(t * BIG_STAGE);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            int ahead = nk - 1 - kt;
            ahead = ahead > BIG_NBUF - 2 ? BIG_NBUF - 2 : ahead;
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // 12 DMA per k-tile per loader wave
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + BIG_NBUF - 1 < nk) {
                int nx = cur + BIG_NBUF - 1;
                nx = nx >= BIG_NBUF ? nx - BIG_NBUF : nx;
                ICKA_BIG_STAGE(nx * BIG_STAGE);
            }
            cur = cur + 1 == BIG_NBUF ? 0 : cur + 1;
        }
#undef ICKA_BIG_STAGE
        return;   // outputs are stored by the compute waves straight from their accumulators
    }
    // ---------------------------------------------------------------------------------------------- compute waves
    const int wsub = wave >> 1;            // which 128-row half of the A tile
    const int wc = (wave & 1) * 64;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 fa0[8], fb0[4], fa1[8], fb1[4];   // fragment sets of the two 32-deep k-steps of a tile
#define ICKA_BIG_READ(FA, FB, BUFI, KS)                                                                     \
    do {                                                                                                    \
        const char* b_ = smem + (BUFI) * BIG_STAGE;                                                         \
        int l_ = lane;                                                                                      \
        asm volatile("" : "+v"(l_)); /* opaque: the 48 swizzled LDS addresses are recomputed per read (a few */ \
        /* VALU ops) instead of being held in registers across the loop -- held, they spilled to scratch and */ \
        /* the serialized reloads cost ~4000 cycles per k-tile */                                           \
        _Pragma("unroll") for (int t = 0; t < 8; ++t) FA[t] = read_frag<true>(b_ + wsub * TILE_BYTES, 16 * t, KS, l_); \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) FB[t] = read_frag<true>(b_ + 2 * TILE_BYTES, wc + 16 * t, KS, l_); \
    } while (0)
#define ICKA_BIG_MMA(FA, FB)                                                                                \
    do {                                                                                                    \
        _Pragma("unroll") for (int mi = 0; mi < 8; ++mi)                                                    \
            _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(FB[ni], FA[mi], acc[mi][ni]); \
    } while (0)
// One scheduling region = 24 fragment reads (into the idle register set) + 32 MFMAs (from the other set): the reads
// are spread between the MFMAs (1 DS read per MFMA, then the remaining MFMAs) instead of being issued as one block
// while the matrix pipe idles -- with one compute wave per SIMD nothing else would fill those cycles.
#define ICKA_BIG_INTERLEAVE()                                                                               \
    do {                                                                                                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 24; ++i_) {                                                 \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                              \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                              \
        }                                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
    __builtin_amdgcn_s_barrier();   // barrier #0: tile 0 published
    asm volatile("" ::: "memory");
    ICKA_BIG_READ(fa0, fb0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    int cur = 0;
    for (int kt = 0; kt + 1 < nk; ++kt) {
        ICKA_BIG_READ(fa1, fb1, cur, 1);
        ICKA_BIG_MMA(fa0, fb0);
        ICKA_BIG_INTERLEAVE();
        const int nxt = cur + 1 == BIG_NBUF ? 0 : cur + 1;
        // tile kt is completely in registers: barrier kt+1 publishes tile kt+1 and frees tile kt's buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ICKA_BIG_READ(fa0, fb0, nxt, 0);
        ICKA_BIG_MMA(fa1, fb1);
        ICKA_BIG_INTERLEAVE();
        cur = nxt;
    }
    ICKA_BIG_READ(fa1, fb1, cur, 1);   // last tile
    ICKA_BIG_MMA(fa0, fb0);
    ICKA_BIG_INTERLEAVE();
    ICKA_BIG_MMA(fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
#undef ICKA_BIG_INTERLEAVE
#undef ICKA_BIG_READ
#undef ICKA_BIG_MMA
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // MFMA -> VALU read wait states (see gemm_ws_body)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc + 16 * ni + 4 * (lane >> 4);
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            const int m = m0 + wsub * 128 + 16 * mi + (lane & 15);
            f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n);
            f32x4 v = acc[mi][ni] * g.alpha;
            if (g.beta != 0.f) v += g.beta * ld_once(dst);   // gradient accumulation across micro-batches
            if (!WIRE || !g.c3_only) st_out(dst, v);
            if constexpr (WIRE) {
                if (g.C3) st_out(reinterpret_cast<u32x2*>(g.C3 + (int64_t)m * g.ldc3 + n), pack4(v[0], v[1], v[2], v[3]));
            }
        }
    }
}

// column sums of a k-major operand slice: colsum[c0 .. c0+64) (+)= sum over the K rows of A[k][m].
// 8 chunk lanes (64 columns) x 64 row lanes, 8 independent 16-byte loads in flight per thread.
__device__ __forceinline__ void big_colsum_block(const GemmArgs& g, char* smem, int cb) {
    float* red = reinterpret_cast<float*>(smem);   // [64 row lanes][64 columns]
    const int tid = threadIdx.x, cx = tid & 7, ry = tid >> 3;
    const int col = cb * CS_COLS + cx * 8;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (col < g.M) {
        const bf16_t* base = g.A + col;
        int k = ry;
        for (; k + 7 * 64 < g.K; k += 8 * 64) {
            u32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(base + (int64_t)(k + 64 * u) * g.lda);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bf16x8 h = as_bf16x8(v[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += bf2f(h[e]);
            }
        }
        for (; k < g.K; k += 64) {
            const bf16x8 h = as_bf16x8(*reinterpret_cast<const u32x4*>(base + (int64_t)k * g.lda));
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += bf2f(h[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ry * CS_COLS + cx * 8 + e] = s[e];
    __syncthreads();
    if (tid < CS_COLS) {
        const int c = cb * CS_COLS + tid;
        if (c < g.M) {
            float t = 0.f;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) t += red[r * CS_COLS + tid];
            g.colsum[c] = g.colsum_acc ? g.colsum[c] + t : t;
        }
    }
}


// 12-wave form of the same tile (8 compute waves of 64 x 64, two per SIMD, + 4 loader waves): fragments are read per
// 32-deep step right before their MFMAs and the second compute wave of the SIMD covers the LDS (ds_read_b64_tr_b16)
// latency, instead of one wave per SIMD with software-pipelined register sets.
template <bool WIRE>
__device__ __forceinline__ void gemm_big12_tn_body(const GemmArgs& g, char* smem, const int m0, const int n0) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>(LDS_PTR(char, smem)));
    const int nk = g.K / BK;
    if (wave >= 8) {
        const int lw = wave - 8;
        const bf16_t* pa0[4];
        const bf16_t* pa1[4];
        const bf16_t* pb[4];
        dma_init<true>(pa0, g.A, g.lda, m0, lw, lane);
        dma_init<true>(pa1, g.A, g.lda, m0 + 128, lw, lane);
        dma_init<true>(pb, g.B, g.ldb, n0, lw, lane);
        const int64_t sa = (int64_t)BK * g.lda, sb = (int64_t)BK * g.ldb;
#define ICKA_BIG_STAGE(BUF)                                              \
    do {                                                                 \
        dma_issue(pa0, sa, lds0 + (BUF) + lw * 1024);                    \
        dma_issue(pa1, sa, lds0 + (BUF) + TILE_BYTES + lw * 1024);       \
        dma_issue(pb, sb, lds0 + (BUF) + 2 * TILE_BYTES + lw * 1024);    \
    } while (0)
#pragma unroll
        for (int t = 0; t < BIG_NBUF - 1; ++t)
            if (t < nk) ICKA_BIG_STAGE(t * BIG_STAGE);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            int ahead = nk - 1 - kt;
            ahead = ahead > BIG_NBUF - 2 ? BIG_NBUF - 2 : ahead;
            if (ahead >= 1) wait_vmcnt<12>();   // 12 DMA per k-tile per loader wave
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + BIG_NBUF - 1 < nk) {
                int nx = cur + BIG_NBUF - 1;
                nx = nx >= BIG_NBUF ? nx - BIG_NBUF : nx;
                ICKA_BIG_STAGE(nx * BIG_STAGE);
            }
            cur = cur + 1 == BIG_NBUF ? 0 : cur + 1;
        }
#undef ICKA_BIG_STAGE
        return;
    }
    const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();   // tile kt published; this wave is done with tile kt-1
        asm volatile("" ::: "memory");
        const char* st = smem + cur * BIG_STAGE;
        const char* sA = st + (wr >> 7) * TILE_BYTES;
        const char* sB = st + 2 * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = read_frag<true>(sA, (wr & 127) + 16 * t, ks, lane);
#pragma unroll
            for (int t = 0; t < 4; ++t) fb[t] = read_frag<true>(sB, wc + 16 * t, ks, lane);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(fb[ni], fa[mi], acc[mi][ni]);
        }
        cur = cur + 1 == BIG_NBUF ? 0 : cur + 1;
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int n = n0 + wc + 16 * ni + 4 * (lane >> 4);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int m = m0 + wr + 16 * mi + (lane & 15);
            f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n);
            f32x4 v = acc[mi][ni] * g.alpha;
            if (g.beta != 0.f) v += g.beta * ld_once(dst);
            if (!WIRE || !g.c3_only) st_out(dst, v);
            if constexpr (WIRE) {
                if (g.C3) st_out(reinterpret_cast<u32x2*>(g.C3 + (int64_t)m * g.ldc3 + n), pack4(v[0], v[1], v[2], v[3]));
            }
        }
    }
}

template <bool W12, bool WIRE = false>   // WIRE: some problem of the group writes the data-parallel wire copy (C3)
__global__ __launch_bounds__(W12 ? 768 : 512) void gemm_big_group_kernel(const BigGroupArgs ga) {
    __shared__ __attribute__((aligned(16))) char smem[BIG_NBUF * BIG_STAGE];
    const int bid = blockIdx.x;
    const int tiles = ga.start[MAX_GROUP];
    if (bid >= tiles + ga.cs_start[MAX_GROUP]) {
        const int rb = bid - tiles - ga.cs_start[MAX_GROUP];
        int ri = 0;
#pragma unroll
        for (int i = 1; i < MAX_RED; ++i) ri += rb >= ga.red_start[i] ? 1 : 0;
        if (W12 && threadIdx.x >= 512) return;   // the helper roles are written for 8 waves
        slab_reduce_block(ga.red[ri], smem, rb - ga.red_start[ri]);
        return;
    }
    if (bid >= tiles) {
        const int cb = bid - tiles;
        int pi = 0;
#pragma unroll
        for (int i = 1; i < MAX_GROUP; ++i) pi += cb >= ga.cs_start[i] ? 1 : 0;
        const GemmArgs g = ga.p[pi];
        if (W12 && threadIdx.x >= 512) return;
        big_colsum_block(g, smem, cb - ga.cs_start[pi]);
        return;
    }
    // Block -> tile across the WHOLE group (blocks b and b+8 share an XCD and its private 4 MiB L2): XCD x takes the
    // contiguous run [x*T/8, (x+1)*T/8) of the problems' concatenated tile lists, each list ordered along its shorter
    // side.  A problem then lives on ~T_p/27 XCDs instead of all 8, each of which has to fetch the operand panels its
    // tiles touch: rocprofv3 FETCH_SIZE was 340 MB per launch against ~100 MB of operands with per-problem striping.
    const int xcd = bid & 7, slot = bid >> 3;
    const int qn = tiles >> 3, rn = tiles & 7;
    const int gt = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + slot;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) pi += gt >= ga.start[i] ? 1 : 0;
    const GemmArgs g = ga.p[pi];
    const int local = gt - ga.start[pi];
    const int nbm = g.M / 256, nbn = g.N / BN;
    int tm, tn;
    if (nbn > nbm) { tm = local % nbm; tn = local / nbm; }
    else { tm = local / nbn; tn = local % nbn; }
    if constexpr (W12) gemm_big12_tn_body<WIRE>(g, smem, tm * 256, tn * BN);
    else gemm_big_tn_body<WIRE>(g, smem, tm * 256, tn * BN);
}

// eligible: fast-path TN, 256-row tiles, plain f32 output (overwrite or accumulate)
static bool big_ok(const GemmArgs& g, bool aligned) {
    return aligned && g.M % 256 == 0 && g.c_f32 && g.epi == ICKA_EPI_NONE && g.K1 == 0 && !g.bias && !g.bias2 && g.direct;
}

template <bool A_KM, bool B_KM>
static int launch_group(const GroupArgs& ga, int total, hipStream_t st, const Tune& t) {
    if (t.ws) {
        hipLaunchKernelGGL((gemm_ws_group_kernel<A_KM, B_KM, 3>), dim3(total), dim3(512), 0, st, ga);
        ICKA_CHECK_LAUNCH();
        return 0;
    }
    const int nbuf = t.nbuf > 0 ? t.nbuf : (total >= 448 ? 2 : 4);
    if (nbuf == 2) hipLaunchKernelGGL((gemm_dma_group_kernel<A_KM, B_KM, 2>), dim3(total), dim3(256), 0, st, ga);
    else if (nbuf == 3) hipLaunchKernelGGL((gemm_dma_group_kernel<A_KM, B_KM, 3>), dim3(total), dim3(256), 0, st, ga);
    else hipLaunchKernelGGL((gemm_dma_group_kernel<A_KM, B_KM, 4>), dim3(total), dim3(256), 0, st, ga);
    ICKA_CHECK_LAUNCH();
    return 0;
}

static int grouped_impl(const icka_gemm_desc* descs, int32_t n, const icka_slab_reduction* reds, int32_t n_red,
                        hipStream_t st);

extern "C" int icka_gemm_grouped(const icka_gemm_desc* descs, int32_t n, void* stream) {
    if (!descs || n <= 0) return ICKA_E_ARG;
    return grouped_impl(descs, n, nullptr, 0, (hipStream_t)stream);
}

extern "C" int icka_gemm_grouped_ex(const icka_gemm_desc* descs, int32_t n, const icka_slab_reduction* reds,
                                    int32_t n_red, void* stream) {
    if (n < 0 || n_red < 0 || (n > 0 && !descs) || (n_red > 0 && !reds) || n_red > MAX_RED) return ICKA_E_ARG;
    for (int r = 0; r < n_red; ++r)
        if (!reds[r].partials || reds[r].nslab <= 0 || reds[r].H <= 0 || reds[r].nslots <= 0 || reds[r].nslots > 4 ||
            reds[r].slab_stride < (int64_t)reds[r].nslots * reds[r].H)
            return ICKA_E_ARG;
    return grouped_impl(descs, n, reds, n_red, (hipStream_t)stream);
}

static SlabRed to_red(const icka_slab_reduction& r) {
    SlabRed o{};
    o.partials = r.partials;
    for (int k = 0; k < 4; ++k) o.out[k] = k < r.nslots ? r.out[k] : nullptr;
    o.nslab = r.nslab; o.H = r.H; o.nslots = r.nslots; o.slab_stride = (int)r.slab_stride; o.accumulate = r.accumulate;
    o.blocks = (r.nslots * r.H + 63) / 64;
    return o;
}

static int grouped_impl(const icka_gemm_desc* descs, int32_t n, const icka_slab_reduction* reds, int32_t n_red,
                        hipStream_t st) {
    bool reds_done = n_red == 0;
    int i = 0;
    // the group's launch heuristics: the environment default, overridden by the FIRST problem's tune word
    // (Tune::big: 256x128 tiles for runs of eligible weight-gradient problems -- 2: 12-wave blocks, +0.9 % on the c2 step)
    Tune t = kTuneEnv;
    if (n > 0 && !tune_of(descs[0].tune, t)) return ICKA_E_ARG;
    Tune tk;   // (per-problem decode: convert() validates every problem's word)
    while (i < n) {
        if (t.big && t.ws && descs[i].op == ICKA_GEMM_TN) {
            // 256x128-tile launch for runs of eligible weight-gradient problems (+ their column-sum blocks)
            BigGroupArgs ba;
            int cnt = 0, total = 0, cs_total = 0;
            while (i + cnt < n && cnt < MAX_GROUP && descs[i + cnt].op == ICKA_GEMM_TN) {
                bool aligned = false;
                GemmArgs g;
                const int rc = convert(&descs[i + cnt], g, aligned, tk);
                if (rc) return rc;
                if (!big_ok(g, aligned)) break;
                ba.p[cnt] = g;
                ba.start[cnt] = total;
                ba.cs_start[cnt] = cs_total;
                total += (g.M / 256) * (g.N / BN);
                if (g.colsum) cs_total += (g.M + CS_COLS - 1) / CS_COLS;
                ++cnt;
            }
            if (cnt >= 1 && total >= 64) {
                for (int k = cnt; k <= MAX_GROUP; ++k) { ba.start[k] = total; ba.cs_start[k] = cs_total; }
                for (int k = cnt; k < MAX_GROUP; ++k) ba.p[k] = ba.p[0];
                int red_total = 0;
                for (int r = 0; r < MAX_RED; ++r) {
                    ba.red_start[r] = red_total;
                    if (!reds_done && r < n_red) { ba.red[r] = to_red(reds[r]); red_total += ba.red[r].blocks; }
                    else ba.red[r] = SlabRed{};
                }
                ba.red_start[MAX_RED] = red_total;
                reds_done = true;
                bool wire = false;
                for (int k = 0; k < cnt; ++k) wire = wire || ba.p[k].C3 != nullptr;
                const dim3 grid(total + cs_total + red_total);
                if (t.big == 2) {
                    if (wire) hipLaunchKernelGGL((gemm_big_group_kernel<true, true>), grid, dim3(768), 0, st, ba);
                    else hipLaunchKernelGGL((gemm_big_group_kernel<true, false>), grid, dim3(768), 0, st, ba);
                } else {
                    if (wire) hipLaunchKernelGGL((gemm_big_group_kernel<false, true>), grid, dim3(512), 0, st, ba);
                    else hipLaunchKernelGGL((gemm_big_group_kernel<false, false>), grid, dim3(512), 0, st, ba);
                }
                ICKA_CHECK_LAUNCH();
                i += cnt;
                continue;
            }
        }
        // greedily pack up to MAX_GROUP consecutive fast-path problems of the same layout into one launch
        GroupArgs ga;
        int cnt = 0, total = 0;
        const int op = descs[i].op;
        while (i + cnt < n && cnt < MAX_GROUP && descs[i + cnt].op == op && !descs[i + cnt].ab_f16) {   // (fp16 problems go alone)
            bool aligned = false;
            const int rc = convert(&descs[i + cnt], ga.p[cnt], aligned, tk);
            if (rc) return rc;
            if (ga.p[cnt].colsum && !(aligned && t.ws)) return ICKA_E_ARG;
            if (!aligned) break;
            ga.start[cnt] = total;
            total += (ga.p[cnt].M / BM) * (ga.p[cnt].N / BN);
            ++cnt;
        }
        if (cnt >= 2) {
            for (int k = cnt; k <= MAX_GROUP; ++k) ga.start[k] = total;
            for (int k = cnt; k < MAX_GROUP; ++k) ga.p[k] = ga.p[0];
            int rc;
            if (op == ICKA_GEMM_TN && !reds_done && t.ws) {
                // the pending slab reductions ride on this launch (extra blocks behind the tiles)
                GroupArgsRed gr;
                gr.ga = ga;
                int red_total = 0;
                for (int r = 0; r < MAX_RED; ++r) {
                    gr.red_start[r] = red_total;
                    if (r < n_red) { gr.red[r] = to_red(reds[r]); red_total += gr.red[r].blocks; }
                    else gr.red[r] = SlabRed{};
                }
                gr.red_start[MAX_RED] = red_total;
                hipLaunchKernelGGL((gemm_ws_group_red_kernel<true, true, 3>), dim3(total + red_total), dim3(512), 0, st, gr);
                ICKA_CHECK_LAUNCH();
                reds_done = true;
                rc = 0;
            } else if (op == ICKA_GEMM_NT) rc = launch_group<false, false>(ga, total, st, t);
            else if (op == ICKA_GEMM_NN) rc = launch_group<false, true>(ga, total, st, t);
            else rc = launch_group<true, true>(ga, total, st, t);
            if (rc) return rc;
            i += cnt;
        } else {
            const int rc = icka_gemm(&descs[i], (void*)st);
            if (rc) return rc;
            ++i;
        }
    }
    if (!reds_done) {   // no 256x128 launch took them along: reduce the slabs with their own small launches
        if (n_red == 1) {
            const SlabRed sr = to_red(reds[0]);
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(sr.blocks), dim3(512), 0, st, sr);
        } else {
            SlabRedMulti m;
            int total = 0;
            for (int r = 0; r < MAX_RED; ++r) {
                m.red_start[r] = total;
                if (r < n_red) { m.red[r] = to_red(reds[r]); total += m.red[r].blocks; }
                else m.red[r] = SlabRed{};
            }
            m.red_start[MAX_RED] = total;
            for (int r = n_red; r < MAX_RED; ++r) m.red_start[r] = total;   // (empty ranges behind the last real one)
            hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(total), dim3(512), 0, st, m);
        }
        ICKA_CHECK_LAUNCH();
    }
    return 0;
}
