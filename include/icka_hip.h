/* icka_hip.h -- C-ABI of libicka_hip.so: the MI355X (gfx950) kernels behind the ICKA MNER hot path.
 *
 * The reference (buctcurry/ICKA) has no FFI / plugin layer: its hot path is PyTorch nn.Module code that lowers to
 * ATen kernels.  Each entry point below therefore names the reference *Python* interface
 * (file:line under /root/reference) whose ATen kernel sequence it replaces.  Plain pointers and sizes only; no
 * torch types.  Every pointer is a DEVICE pointer unless stated; `stream` is a hipStream_t passed as void*.
 * bf16 tensors are passed as `const void*` to 2-byte elements; "f32" means float.
 *
 * Return value: 0 on success, a positive hipError_t from the launch, or a negative ICKA_E_* argument error.
 * Nothing here allocates, frees or synchronises: every call is stream-ordered and hipGraph-capturable.
 *
 * Re-entrancy: the compute entry points read NO mutable process state.  What varies between launches of
 * one shape -- the alternative kernels the tests exercise, A/B runs of a heuristic -- is an explicit per-call argument
 * (icka_gemm_desc.tune, the `flags` of icka_attn_* / icka_lstm_*) or an ICKA_TUNE_* environment variable read ONCE when the
 * library is loaded; there are no tuning setters.  The only process-wide controls left are documented where they are declared:
 * the dropout nonce registration (icka_set_dropout_nonce), the error words of the persistent kernels
 * (icka_lstm_clear_error, icka_dp_clear_error), the CU reservation beside RCCL (icka_lstm_set_reserved_cus) and the test hooks
 * of the give-up paths (icka_lstm_test_hooks, icka_gemm_ln_test_hooks).
 */
#ifndef ICKA_HIP_H
#define ICKA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICKA_E_SHAPE (-1)   /* dimension out of the supported range */
#define ICKA_E_ALIGN (-2)   /* pointer / leading dimension not 16-byte aligned where vector access needs it */
#define ICKA_E_ARG (-3)     /* null pointer or inconsistent descriptor */

/* library / ABI version, and the code-object architecture it was built for ("gfx950") */
/* 2: round 2 -- icka_gemm_desc grew (ab_f16, C3, ldc3, aux_f16), c_is_f32 may be 2 (fp16 output); new entry points
 * icka_ln_fwd_h, icka_embed_fwd_h, icka_attn_fwd_ex, icka_cls_head_fwd_h, icka_cast_*f16, icka_conv3x3_gemm.
 * 3: icka_lstm_set_handoff, icka_lstm_set_batch_split, icka_attn_dropout_mask (additive).
 * 4: round 3 -- icka_lstm_clear_error, icka_lstm_set_reserved_cus, icka_lstm_test_hooks; a hand-off wait that gives up now
 *    NaN-poisons the recurrence and raises a host-visible error word; icka_gemm_desc.C3 may accompany an f32 main output
 *    (the data-parallel wire copy, c3_only); icka_dp_*; icka_regions_to_tokens_h, icka_sample_gate_fwd_h, icka_optim_* (additive).
 * 5: round 4 -- icka_dp_flag_wait's third argument is the bucket's BAD WORD (it no longer writes the NaN itself);
 *    icka_dp_poison_if, icka_dp_poison_final, icka_copy_many, icka_embed_bwd_rows, icka_embed_scatter_rows,
 *    icka_ln_set_rows_per_wave, icka_gemm_set_square_tiles, icka_gemm_set_persistent (additive); icka_gemm refuses an f32 output with a wire copy (C3) unless op == TN.
 * 6: round 5 -- every tuning setter is GONE (icka_gemm_set_*, icka_ln_set_rows_per_wave, icka_attn_set_whole_head,
 *    icka_lstm_set_persistent / _handoff / _batch_split): icka_gemm_desc grew `tune`; icka_attn_fwd_ex's `fp8` argument became
 *    `flags`; icka_attn_bwd, icka_lstm_fwd and icka_lstm_bwd take `flags`; the 256x256-tile and persistent 12-wave GEMM kernels
 *    those setters switched on left the library (profiles/NEGATIVE_RESULTS.md); icka_gemm_ln, icka_gemm_ln_sync_words, icka_gemm_ln_test_hooks, icka_gemm_qkv_attn (additive). */
#define ICKA_ABI_VERSION 6
int icka_abi_version(void);
const char* icka_build_arch(void);

/* n <= 8 device-to-device copies (dst[i] <- src[i], bytes[i] bytes; src / dst / bytes are HOST arrays) in one launch: the
 * per-call input refresh of a captured step (icka_amd/graph.py: StaticInputs), which stands where the reference's loop
 * moves a new batch to the device every step (My_cross_attention.py:797-798).  16-byte words when pointers and size allow. */
int icka_copy_many(const void* const* src, void* const* dst, const int64_t* bytes, int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * GEMM on MFMA (v_mfma_f32_16x16x32_bf16), bf16 operands, fp32 accumulate.
 *   op NT :  C[M,N] = A[M,K]   . B[N,K]^T      (nn.Linear forward: x @ W^T, W is [out,in])
 *   op NN :  C[M,N] = A[M,K]   . B[K,N]        (input gradient: dY @ W)
 *   op TN :  C[M,N] = A[K,M]^T . B[K,N]        (weight gradient: dY^T @ X, K = tokens)
 * Replaces nn.Linear / torch.matmul call sites: Cross_Modal_Interaction_Module.py:479-481 (Q,K,V), :533,:549,:562
 * (dense layers), :958 (vismap2text), my_bert/cl_modeling.py:1363,:1371 (gates, classifier) and their autograd.
 * The reduction may be split in two segments read from different buffers (k < K1 from A/B, k >= K1 from A2/B2):
 * that is how cat(seq, cross) operands (cl_modeling.py:1363-1370) are consumed without materialising the concat.
 */
enum { ICKA_GEMM_NT = 0, ICKA_GEMM_NN = 1, ICKA_GEMM_TN = 2 };
enum {
    ICKA_EPI_NONE = 0,  /* C = alpha*acc (+bias) (+beta*C)                                                  */
    ICKA_EPI_GELU = 1,  /* z = acc+bias ; C2 = z ; C = z*0.5*(1+erf(z/sqrt2))   (BertIntermediate, :548-551)   */
    ICKA_EPI_DGELU = 2, /* C = acc * gelu'(aux)                                  (its backward)                */
    ICKA_EPI_ADD = 3,   /* C = acc (+bias) + aux                                 (gradient fan-in)             */
    ICKA_EPI_GATE = 4,  /* g = sigmoid(acc+bias) ; C2 = g ; C = g*aux            (cl_modeling.py:1363-1367)    */
    ICKA_EPI_TANH = 5,  /* C = tanh(acc+bias)                                    (BertPooler, :675-681)        */
    ICKA_EPI_RELU = 6,  /* C = max(acc+bias, 0)                 (conv+BN+ReLU, resnet/resnet.py:75-81)          */
    ICKA_EPI_ADD_RELU = 7 /* C = max(acc+bias+aux, 0)           (bottleneck residual, resnet/resnet.py:83-90)   */
};
typedef struct icka_gemm_desc {
    int32_t op;            /* ICKA_GEMM_* */
    int32_t M, N, K;
    int32_t K1;            /* 0 = single segment; else multiple of 64, 0 < K1 < K */
    const void* A;  int64_t lda;    /* bf16 */
    const void* B;  int64_t ldb;    /* bf16 */
    const void* A2; int64_t lda2;   /* second-segment operands (K1 > 0) */
    const void* B2; int64_t ldb2;
    void* C;  int64_t ldc;  int32_t c_is_f32;   /* output bf16 (0), f32 (1) or fp16 (2, see ab_f16 below) */
    void* C2; int64_t ldc2;                     /* bf16 second output (GELU: z, GATE: g) */
    const void* aux; int64_t ldaux;             /* bf16 epilogue operand */
    const float* bias;                          /* f32 [N] or NULL */
    const float* bias2;                         /* optional second f32 [N] bias (two Linear layers summed) */
    float alpha, beta;
    int32_t epilogue;      /* ICKA_EPI_* */
    float* colsum_out;     /* op TN only, fast path (M,N %128, K %64): colsum_out[m] (+)= sum_k A[k,m]  -- the bias
                              gradient db = colsum(dY) produced by the weight-gradient GEMM dW = dY^T.X itself */
    int32_t colsum_accumulate;
    /* "mixed16" forward GEMMs (op NT only): operands A, B (A2, B2) are IEEE fp16 instead of bf16
     * (v_mfma_f32_16x16x32_f16, same rate); c_is_f32 == 2 makes the main output fp16 (saturating at +-65504, beta must
     * be 0) and C3, if not NULL, receives a bf16 copy of it (the operand the bf16 weight-gradient GEMM reads later).
     * With an f32 main output (c_is_f32 == 1, op TN ONLY: other ops return ICKA_E_ARG) C3 is the DATA-PARALLEL WIRE COPY: a bf16 copy of the final value
     * (after beta-accumulation) written by the same epilogue -- the weight-gradient GEMMs fill the bf16 all-reduce buffer
     * of icka_amd/dp.py themselves instead of a separate cast pass over the gradients (see icka_dp_* below). */
    int32_t ab_f16;
    void* C3; int64_t ldc3;
    int32_t aux_f16;       /* the epilogue operand `aux` is fp16 instead of bf16 (mixed16 gate: sigmoid(.) * cross_fp16) */
    int32_t c3_only;       /* f32 C with a wire copy C3, beta == 0, no epilogue: store ONLY C3 (2 bytes per element instead of
                              4 + 2); C is left untouched -- icka_amd/dp.py writes it from the reduced wire buffer
                              (icka_dp_cast_back_scaled), which is the only consumer of such a gradient before that point */
    uint64_t tune;         /* 0 = the library's per-shape heuristics (the ICKA_TUNE_GEMM_* environment read at load, if set);
                              else an OR of ICKA_TUNE_* words that force single choices for THIS call (tests of the alternative
                              kernels, probes).  Results never depend on it; a code outside a field's range is ICKA_E_ARG.
                              A grouped launch takes the word of its first problem. */
} icka_gemm_desc;
/* icka_gemm_desc.tune fields (4 bits each; environment equivalents in parentheses):
 *   RING(n)        LDS-DMA ring depth of the fast path, n = 2 .. 5 (ICKA_TUNE_GEMM_RING): 2 = 64 KiB, two blocks per CU; 3 / 4 =
 *                  96 / 128 KiB, one block per CU, one / two k-tiles of DMA kept in flight across the barrier.  Default: ring 2
 *                  with two co-resident blocks for short-K grids of >= ~2 tiles per CU, ring 3 otherwise.
 *   TILE_N(n)      output-tile width of the warp-specialised path, n = 96 | 128 (ICKA_TUNE_GEMM_TILE_N).  Default: the narrower
 *                  tile when it quantises better onto the 256 CUs (N = 768: 256 tiles instead of 192); 96 needs N % 96 == 0.
 *   WIDE_TILES(b)  256x192 / 256x128 tiles of the 12-wave kernel for wide outputs with a short reduction whose tile grid
 *                  covers the CUs in one round (ICKA_TUNE_GEMM_WIDE_TILES; default on).
 *   DIRECT_EPILOGUE(b)  f32 outputs without activation / fan-in operand / accumulate stored straight from the MFMA
 *                  accumulators (default on); off: every epilogue goes through the LDS C tile (ICKA_TUNE_GEMM_DIRECT_EPILOGUE).
 *   WARP_SPECIALIZED(v)  1 (default): 512-thread fast path (4 loader + 4 compute waves; two blocks per CU for large grids),
 *                  0: 256-thread single-role path, 2: force two blocks, 3: never two (ICKA_TUNE_GEMM_WARP_SPECIALIZED).
 *   W3_GRID(pm)    XCD cut of the 12-wave kernel's tile grid, pm = 1 | 2 | 4 | 8 where it divides the grid: the 8 XCDs (private
 *                  L2s) take a pm x pn patch grid of the tiles, the chip then fetches pn * |A| + pm * |B|
 *                  (ICKA_TUNE_GEMM_W3_GRID).  Default: the dividing cut with the smallest pn * M + pm * N.
 *   BIG_TILES(v)   grouped TN launches of plain f32 outputs with M % 256 == 0 (the weight gradients of a layer): 2 (default)
 *                  = 256x128 tiles in 12-wave blocks with the fused column sums in extra blocks of the grid, 1 = 8-wave
 *                  blocks, 0 = 128x128 tiles (ICKA_TUNE_GEMM_BIG_TILES).
 *   ABLATION(m)    diagnostic builds only, WRONG results: 1 = skip MFMA + LDS reads, 2 = skip the LDS-DMA staging. */
#define ICKA_TUNE_RING(n) ((uint64_t)(n) << 0)
#define ICKA_TUNE_TILE_N(n) ((uint64_t)((n) == 96 ? 1 : ((n) == 128 ? 2 : 15)) << 4)
#define ICKA_TUNE_WIDE_TILES(b) ((uint64_t)((b) ? 2 : 1) << 8)
#define ICKA_TUNE_DIRECT_EPILOGUE(b) ((uint64_t)((b) ? 2 : 1) << 12)
#define ICKA_TUNE_WARP_SPECIALIZED(v) ((uint64_t)((v) + 1) << 16)
#define ICKA_TUNE_W3_GRID(pm) ((uint64_t)(pm) << 20)
#define ICKA_TUNE_BIG_TILES(v) ((uint64_t)((v) + 1) << 24)
#define ICKA_TUNE_ABLATION(m) ((uint64_t)(m) << 28)
int icka_gemm(const icka_gemm_desc* d, void* stream);
/* n independent GEMMs; consecutive fast-path problems of one layout are packed (up to 4) into ONE launch so that
 * several partially-filling grids (the weight-gradient GEMMs of a layer) fill the chip together. */
int icka_gemm_grouped(const icka_gemm_desc* descs, int32_t n, void* stream);
/* Same, plus up to 4 slab reductions  out[slot][c] (+)= sum_b partials[b*slab_stride + slot*H + c]  (b < nslab,
 * slot < nslots <= 4, c < H; fixed summation order) that ride on the launch: the LayerNorm dgamma / dbeta slabs left
 * by icka_ln_bwd_slabs are summed by extra blocks of the layer's weight-gradient grid instead of a launch of their
 * own.  A 128x128 TN group takes them along the same way; without any eligible launch (or with n == 0) they share ONE small launch. */
typedef struct icka_slab_reduction {
    const float* partials; int64_t slab_stride; int32_t nslab, H, nslots, accumulate;
    float* out[4];
} icka_slab_reduction;
int icka_gemm_grouped_ex(const icka_gemm_desc* descs, int32_t n, const icka_slab_reduction* reds, int32_t n_red,
                         void* stream);
/* ---------------------------------------------------------------------------------------------------------------
 * Fused  y = LayerNorm(dropout(x + bias) + residual)   (BertSelfOutput.forward :561-565, BertOutput.forward
 * :532-536, BertLayerNorm.forward :518-522: biased variance, eps inside the sqrt).  One wave per row.
 *   x [M,H] bf16 or f32 (x_is_f32; row stride ldx in elements), bias f32[H] or NULL, residual [M,H] bf16 or f32
 *   (res_is_f32; ldr) or NULL, gamma/beta f32[H];  y [M,H] bf16 (ldy) is the MFMA operand of the next GEMM; y2
 *   optional second bf16 copy (ldy2); y_f32 optional contiguous f32 copy (the residual stream is carried in f32
 *   so bf16 rounding does not accumulate over layers); xhat [M,H] bf16 contiguous and rstd f32[M] are the saved
 *   statistics for backward (may be NULL in inference).
 * A forward wave owns two rows from 4096 rows on (the loads of its next row are issued before the current row is reduced and
 *   stored), else one; ICKA_TUNE_LN_ROWS_PER_WAVE = 1 .. 16 (read once at load) replaces that choice for A/B runs.
 */
int icka_ln_fwd(const void* x, int64_t ldx, int32_t x_is_f32, const float* bias, const void* residual, int64_t ldr,
                int32_t res_is_f32, const float* gamma, const float* beta, void* y, int64_t ldy, void* y2,
                int64_t ldy2, float* y_f32, void* xhat, float* rstd, int32_t M, int32_t H, float eps, float p_drop,
                uint64_t seed, void* stream);
/* dense -> bias + dropout + residual -> LayerNorm as ONE launch (BertSelfOutput.forward :561-565, BertOutput.forward :532-536 =
 * icka_gemm + icka_ln_fwd with the launch boundary replaced by a per-stripe arrival counter inside the kernel).  `d` describes the
 * dense GEMM exactly as icka_gemm would take it for this site: op NT, bf16 operands, C = the f32 intermediate [M, N] (written,
 * then read back by the LayerNorm phase), no bias / epilogue / accumulate in `d` (bias is the LayerNorm phase's, as in
 * icka_ln_fwd).  The remaining arguments are icka_ln_fwd's (x = d->C): residual [M, N] of kind res_kind (0 bf16, 1 f32,
 * 2 fp16; ldr), gamma / beta f32 [N], y bf16 [M, N] (ldy), y_twin contiguous f32 (twin_f16 = 0) or fp16 (1) or NULL, xhat
 * bf16 contiguous and rstd f32 [M] or NULL.  Results are BITWISE those of the two calls.
 * Eligible shapes: the aligned fast path (M % 128 == 0) with eight column tiles per 128-row stripe (N = 768 or 1024) and at most
 * one block per CU (the 8 blocks of a stripe wait for each other: all must be resident; M <= 4096 on a whole MI355X).  Anything
 * else returns ICKA_E_SHAPE and launches nothing: the caller takes the two calls.
 * and M / 128 * 8 blocks <= the device's CUs MINUS the caller's reserve (icka_lstm_set_reserved_cus: dp.GradReducer reserves
 * the CUs RCCL's workgroups may hold, so the fused form is not taken beside a collective).
 * sync_words: icka_gemm_ln_sync_words() 32-bit words of device memory, zero before the first use; every launch leaves them
 * zero again (the counters reset themselves, also after a failed launch), so one buffer serves all fused launches of a stream.
 * error_word: one 32-bit word the device can write -- HOST-MAPPED memory if the host wants to poll it without a device
 * synchronisation (icka_amd/kernels.py keeps a pinned word).  A stripe wait that gave up (bounded spin, ~7 ms: never a hang)
 * stores 1 there and turns the rows it finished into NaN: a failed launch never passes for a result.
 * icka_gemm_ln_test_hooks (tests of that path): polls > 0 replaces the poll budget (0 restores it); drop_block >= 0 makes
 * that block never arrive, so its stripe's waits give up (-1: off). */
int icka_gemm_ln(const icka_gemm_desc* d, const float* bias, const void* residual, int64_t ldr, int32_t res_kind,
                 const float* gamma, const float* beta, void* y, int64_t ldy, void* y_twin, int32_t twin_f16, void* xhat,
                 float* rstd, float eps, float p_drop, uint64_t seed, uint32_t* sync_words, uint32_t* error_word, void* stream);
int64_t icka_gemm_ln_sync_words(void);
int icka_gemm_ln_test_hooks(int32_t polls, int32_t drop_block);
/* The fused QKV projection and the self-attention of BertSelfAttention.forward (Cross_Modal_Interaction_Module.py:478-506:
 * query / key / value Linear, transpose_for_scores, scores / sqrt(d) + mask, softmax, dropout, context) as ONE launch, for
 * sequences of 128 tokens (the reference's max_seq_length) or 256 (BASELINE config c4) and head size 64.  `d` is the projection
 * exactly as icka_gemm takes it: op NT, A = hidden states [B * S, K], B = the stacked [Wq; Wk; Wv] [3 H, K] (both bf16, or both
 * fp16: the "mixed16" forward operands), bias = the stacked f32 bias, C = the bf16 [B * S, 3 H] activation [q | k | v] (WRITTEN
 * as by icka_gemm: the backward reads it), no epilogue / accumulate / copies.  The rest is icka_attn_fwd_ex's with Q / K / V = the
 * three column blocks of C: add_mask f32 [B, S], ctx bf16 [B * S, H] (ldo) and its optional fp16 copy ctx_f16 (same ldo), lse
 * f32 [B, heads, S] or NULL, scale, p_drop, seed (same dropout counters: icka_attn_bwd regenerates the same mask), keep_bits
 * (icka_attn_keepbits_words words or NULL; S = 256 only).
 * Each 256 x 192 tile of the 12-wave GEMM kernel is laid over 256 / S samples x ONE head's 64 + 64 + 64 columns, so the block
 * that produced q, k, v of a head runs that head's attention from LDS.  Results are BITWISE those of icka_gemm +
 * icka_attn_fwd_ex.  Eligible: S == 128 or 256, H = 64 * heads, B * S % 256 == 0, K % 64 == 0 and <= 1024, (B * S / 256) * heads
 * a multiple of 8 and >= 128; anything else returns ICKA_E_SHAPE and launches nothing: the caller makes the two calls. */
int icka_gemm_qkv_attn(const icka_gemm_desc* d, const float* add_mask, void* ctx, void* ctx_f16, int64_t ldo, float* lse,
                       int32_t B, int32_t heads, int32_t S, float scale, float p_drop, uint64_t seed, void* keep_bits, void* stream);
/* "mixed16" form of the same call: x_kind / res_kind are 0 = bf16, 1 = f32, 2 = fp16, and the twin copy of the output is
 * fp16 (y_f16, contiguous, saturating at +-65504): it is both the fp16 MFMA operand of the next forward GEMM and the
 * residual input of the next block, while y (bf16) stays the operand of the bf16 weight-gradient GEMM in backward. */
int icka_ln_fwd_h(const void* x, int64_t ldx, int32_t x_kind, const float* bias, const void* residual, int64_t ldr,
                  int32_t res_kind, const float* gamma, const float* beta, void* y, int64_t ldy, void* y2,
                  int64_t ldy2, void* y_f16, void* xhat, float* rstd, int32_t M, int32_t H, float eps, float p_drop,
                  uint64_t seed, void* stream);
/* Backward of the above.  dy (+ optional dy2) are the incoming gradients of y.  Outputs: dres = gradient of the
 * residual input (bf16, may be NULL), dx = gradient of x (dropout mask re-generated from seed; may be NULL),
 * and the f32 parameter gradients dgamma[H], dbeta[H], dbias[H] (dbias may be NULL): added to the existing values
 * when accumulate != 0, overwritten otherwise.  `partials` is f32 workspace of icka_ln_bwd_workspace_floats(H). */
int64_t icka_ln_bwd_workspace_floats(int32_t H);
int icka_ln_bwd(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* xhat, const float* rstd,
                const float* gamma, void* dres, int64_t lddres, void* dx, int64_t lddx, float* dgamma, float* dbeta,
                float* dbias, float* partials, int32_t M, int32_t H, float p_drop, uint64_t seed, int32_t accumulate,
                void* stream);
/* The same backward without the parameter-gradient finalize: only dres / dx and the column slabs
 * partials[icka_ln_bwd_nslab(M)][icka_ln_slab_slots()][H] (slot 0: sum dy*xhat -> dgamma, slot 1: sum dy -> dbeta),
 * to be summed later by icka_gemm_grouped_ex. */
int icka_ln_bwd_slabs(const void* dy, int64_t lddy, const void* dy2, int64_t lddy2, const void* xhat, const float* rstd,
                      const float* gamma, void* dres, int64_t lddres, void* dx, int64_t lddx, float* partials, int32_t M,
                      int32_t H, float p_drop, uint64_t seed, void* stream);
int32_t icka_ln_bwd_nslab(int32_t M);
int32_t icka_ln_slab_slots(void);

/* ---------------------------------------------------------------------------------------------------------------
 * BertEmbeddings.forward (:398-412): y = dropout(LayerNorm(word[ids] + pos[arange(S)] + type[tt])).
 * Tables are the fp32 master parameters (gathered directly, no shadow copy).  ids/tt are int64 [B*S].
 */
int icka_embed_fwd(const int64_t* ids, const int64_t* token_type, const float* word, const float* pos,
                   const float* type, const float* gamma, const float* beta, void* y, float* y_f32, void* xhat,
                   float* rstd, int32_t B, int32_t S, int32_t H, int32_t vocab, int32_t n_type, float eps,
                   float p_drop, uint64_t seed, void* stream);
/* "mixed16" form: the twin copy of the output is fp16 (see icka_ln_fwd_h). */
int icka_embed_fwd_h(const int64_t* ids, const int64_t* token_type, const float* word, const float* pos,
                     const float* type, const float* gamma, const float* beta, void* y, void* y_f16, void* xhat,
                     float* rstd, int32_t B, int32_t S, int32_t H, int32_t vocab, int32_t n_type, float eps,
                     float p_drop, uint64_t seed, void* stream);
/* Backward.  dword / dpos are ALWAYS accumulated into with f32 atomics (the caller zeroes them for a fresh
 * gradient); dtype / dgamma / dbeta follow `accumulate` as in icka_ln_bwd.  Row `padding_idx` of the word table
 * receives no gradient (nn.Embedding(padding_idx=0), :387).  partials: icka_ln_bwd_workspace_floats(H) floats. */
int icka_embed_bwd(const void* dy, const int64_t* ids, const int64_t* token_type, const void* xhat,
                   const float* rstd, const float* gamma, float* dword, float* dpos, float* dtype, float* dgamma,
                   float* dbeta, float* partials, int32_t B, int32_t S, int32_t H, int32_t vocab, int32_t n_type,
                   int32_t padding_idx, float p_drop, uint64_t seed, int32_t accumulate, void* stream);
/* Row-sparse form for the data-parallel exchange of the word-embedding gradient (icka_amd/dp.py: GradReducer(sparse_embeddings=
 * True); reference: apex DDP all-reduces the dense [vocab, H] gradient, My_cross_attention.py:768-776): as icka_embed_bwd, but
 * the word-table gradient is left as per-token rows dtok f32 [B*S, H] (zero rows for padding_idx).  After an all-gather of every
 * rank's rows and ids, icka_embed_scatter_rows adds scale * rows[t] into dword[ids[t]] (f32 atomics; rows f32 or bf16; the
 * padding id and ids outside [0, vocab) are skipped; the caller zeroes dword for a fresh gradient). */
int icka_embed_bwd_rows(const void* dy, const int64_t* ids, const int64_t* token_type, const void* xhat, const float* rstd,
                        const float* gamma, float* dtok, float* dpos, float* dtype, float* dgamma, float* dbeta, float* partials,
                        int32_t B, int32_t S, int32_t H, int32_t vocab, int32_t n_type, int32_t padding_idx, float p_drop,
                        uint64_t seed, int32_t accumulate, void* stream);
int icka_embed_scatter_rows(const void* rows, int32_t rows_are_bf16, const int64_t* ids, float* dword, int64_t T, int32_t H,
                            int32_t vocab, int32_t padding_idx, float scale, void* stream);
/* Embeddings of the prompt-accepting encoder stage that ends the current reference model
 * (Cross_Modal_Interaction_Module.py:1010-1012: last_encoder(input_ids=..., prompt_embeddings=prefix_emb, ...); its
 * package `local_transformers` is absent from the reference tree, so the splice rule is this build's definition, taken
 * from the reference's own offset arithmetic :1022 and the token dump at My_cross_attention.py:402-404):
 *   out[b, t] = dropout(LayerNorm(x + pos[t + pos_offset] + type[0])),   t in [0, S)
 *   x = word[ids[b, src[t]]]            if src[t] >= 0   (ids int64 [B, S_in])
 *     = prompt[b, -1 - src[t]]          otherwise        (prompt bf16 [B, P, H])
 * src is int32 [S], shared by the batch (the reference asserts equal offsets per batch, My_cross_attention.py:802). */
int icka_embed_prompt_fwd(const int64_t* ids, const int32_t* src, const void* prompt, const float* word,
                          const float* pos, const float* type, const float* gamma, const float* beta, void* y,
                          float* y_f32, void* xhat, float* rstd, int32_t B, int32_t S_in, int32_t S, int32_t P,
                          int32_t H, int32_t vocab, int32_t pos_offset, float eps, float p_drop, uint64_t seed,
                          void* stream);
/* Backward: table gradients as icka_embed_bwd (dtype = row 0 only), plus dprompt bf16 [B, P, H] (each row written
 * once).  S <= 1024.  partials: S * icka_ln_slab_slots() * H floats. */
int icka_embed_prompt_bwd(const void* dy, const int64_t* ids, const int32_t* src, const void* xhat, const float* rstd,
                          const float* gamma, float* dword, float* dpos, float* dtype, float* dgamma, float* dbeta,
                          void* dprompt, float* partials, int32_t B, int32_t S_in, int32_t S, int32_t P, int32_t H,
                          int32_t vocab, int32_t pos_offset, int32_t padding_idx, float p_drop, uint64_t seed,
                          int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Fused multi-head attention, head size 64 (bert-base 768/12, bert-large 1024/16).
 * BertSelfAttention.forward (:478-506) and BertCoAttention.forward (:590-624):
 *   scores = Q.K^T ; scores *= scale ; scores += add_mask[b, key] ; P = softmax ; P = dropout(P) ; O = P.V
 * Q/K/V/O are bf16 token-major matrices: element (b, s, head, e) at  ptr[(b*S + s)*ld + head*64 + e], so the
 * fused [M,3H] QKV projection output is consumed in place and O is written head-merged ([M,H], :503-505).
 * add_mask f32 [B,Skv] is the additive mask (1-mask)*-10000 (:364-372, :962-965).  lse f32 [B,heads,Sq] is saved
 * for the backward, which recomputes P instead of storing the [B,h,Sq,Skv] probabilities.
 */
int icka_attn_fwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  const float* add_mask, void* O, int64_t ldo, float* lse, int32_t B, int32_t heads, int32_t Sq,
                  int32_t Skv, float scale, float p_drop, uint64_t seed, void* stream);
/* BASELINE config c5: the same forward with QK^T and PV on the fp8 matrix cores (OCP e4m3, fp32 accumulate; P is
 * scaled by 2^8 into e4m3's normal range, softmax statistics stay fp32).  Whole-head shapes only (Sq, Skv <= 128: the
 * text->image cross-attention); the backward is icka_attn_bwd (bf16 recomputation from the saved lse). */
int icka_attn_fwd_fp8(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                      const float* add_mask, void* O, int64_t ldo, float* lse, int32_t B, int32_t heads, int32_t Sq,
                      int32_t Skv, float scale, float p_drop, uint64_t seed, void* stream);
/* icka_attn_fwd / icka_attn_fwd_fp8 (fp8 != 0) with an additional fp16 copy O_f16 (may be NULL; same leading dimension
 * ldo) of the context: the "mixed16" mode feeds it to the out-proj GEMM as fp16 operand, O (bf16) stays the operand of
 * the bf16 weight-gradient GEMM and of icka_attn_bwd.
 * flags: ICKA_ATTN_FP8 = the fp8 QK^T / PV form (icka_attn_fwd_fp8); ICKA_ATTN_TILED = take the tiled flash-style kernels
 * also where the head would fit the whole-head kernels below (per call; the tests exercise both paths).
 * keep_bits (may be NULL; used when p_drop > 0): icka_attn_keepbits_words(B, heads, Sq, Skv) 32-bit words that receive the
 * keep decisions of the attention-probability dropout (nn.Dropout on the probabilities, :500 / :616) -- the same decisions
 * icka_attn_dropout_mask materialises -- so that icka_attn_bwd, given the same buffer, reads bits instead of hashing every
 * element again (about half of the backward's vector work).  Layout: per (batch*head, query, g = (key % 16) / 4)
 * ceil(Skv / 128) words, bit (key / 16) * 4 + key % 4.  Round 3: ABI version 4 added this parameter. */
int icka_attn_fwd_ex(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                     const float* add_mask, void* O, void* O_f16, int64_t ldo, float* lse, int32_t B, int32_t heads,
                     int32_t Sq, int32_t Skv, float scale, float p_drop, uint64_t seed, int32_t flags, void* keep_bits,
                     void* stream);
#define ICKA_ATTN_FP8 1
#define ICKA_ATTN_TILED 2
int64_t icka_attn_keepbits_words(int32_t B, int32_t heads, int32_t Sq, int32_t Skv);
/* Heads with Sq <= 128 and Skv <= 128 (the reference's max_seq_length 128 and 36/49 regions) take the whole-head
 * kernels: one block per (batch, head), forward without online-softmax rescaling, backward (dQ, dK, dV, delta) in
 * one launch (ICKA_ATTN_TILED in a call's flags takes the tiled kernels instead). */
/* delta f32 [B,heads,Sq] is workspace (rowsum(dO*O) = rowsum(P.dP)); it is written by the call.  keep_bits: NULL, or the
 * buffer the forward (icka_attn_fwd_ex) filled for the same shape, seed and replay nonce (whole-head kernels read it, the tiled
 * ones hash). */
int icka_attn_bwd(const void* Q, int64_t ldq, const void* K, int64_t ldk, const void* V, int64_t ldv,
                  const float* add_mask, const void* O, int64_t ldo, const void* dO, int64_t lddo, const float* lse,
                  float* delta, void* dQ, int64_t lddq, void* dK, int64_t lddk, void* dV, int64_t lddv, int32_t B,
                  int32_t heads, int32_t Sq, int32_t Skv, float scale, float p_drop, uint64_t seed, const void* keep_bits,
                  int32_t flags, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Element-wise / layout helpers.
 */
/* fp32 -> bf16 cast of a flat buffer (parameter shadow refresh); n elements. */
int icka_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
int icka_cast_bf16_to_f32(const void* src, float* dst, int64_t n, void* stream);
/* both 16-bit weight shadows from the f32 masters in one pass: bf16 (backward operands) and fp16 (saturating; the
 * forward operands of the "mixed16" mode).  16-byte aligned pointers. */
/* contiguous bf16 (src_is_f32 = 0) or f32 (1) -> fp16, saturating at +-65504 */
int icka_cast_to_f16(const void* src, int32_t src_is_f32, void* dst, int64_t n, void* stream);
int icka_cast_f32_to_bf16_f16(const float* src, void* dst_bf16, void* dst_f16, int64_t n, void* stream);
/* 2-D cast with zero padding: dst bf16 [M, ldd] ; dst[:, :N] = src f32 [M,N] (row stride lds), dst[:, N:ldd] = 0.
 * (logit gradients [M,C] -> a 16-byte-aligned bf16 GEMM operand.) */
int icka_cast_pad_f32_to_bf16(const float* src, int64_t lds, void* dst, int64_t ldd, int32_t M, int32_t N,
                              void* stream);
/* int64 0/1 mask [B, T] (row stride ld, first T columns used) -> additive f32 [B,T] = (1-m)*-10000  (:364-372). */
int icka_additive_mask(const int64_t* mask, int64_t ld, float* out, int32_t B, int32_t T, void* stream);
/* y = dropout(x) with the counter-hash mask; the same call with the same seed is its own backward (nn.Dropout,
 * Cross_Modal_Interaction_Module.py:953).  x,y bf16, row strides ldx/ldy; y2 optional second copy. */
int icka_dropout(const void* x, int64_t ldx, void* y, int64_t ldy, void* y2, int64_t ldy2, int32_t M, int32_t H,
                 float p_drop, uint64_t seed, void* stream);
/* Region features -> bf16 region tokens [B*R, C]:  layout 0: src f32 [B,R,C];  layout 1: src f32 [B,C,R]
 * (myResnet 'att' output [B,2048,7,7] viewed [B,2048,49] and permuted, :956). */
int icka_regions_to_tokens(const float* src, void* dst, int32_t B, int32_t R, int32_t C, int32_t layout,
                           void* stream);
/* "mixed16": the same with an fp16 twin of the tokens written in the same pass (dst_f16, same shape). */
int icka_regions_to_tokens_h(const float* src, void* dst_bf16, void* dst_f16, int32_t B, int32_t R, int32_t C, int32_t layout,
                             void* stream);
/* out[N] (+)= column sums of x[M,N] bf16 (bias gradients). */
int icka_colsum(const void* x, int64_t ldx, float* out, float* partials, int32_t M, int32_t N, int32_t accumulate,
                void* stream);
int64_t icka_colsum_workspace_floats(int32_t N);
/* Gate backward (cl_modeling.py:1363-1367): given dout = d(g*cross), g, cross:
 *   du = dout*cross*g*(1-g) ;  dcross = dout*g (+ dcross_in if not NULL).  All bf16 [M,H]. */
int icka_gate_bwd(const void* dout, int64_t lddout, const void* g, const void* cross, int64_t ldcross,
                  const void* dcross_in, int64_t lddci, void* du, void* dcross, int64_t lddc, int32_t M, int32_t H,
                  void* stream);
/* Classifier of the gated head, C <= 16 labels (cl_modeling.py:1371 `logits = self.classifier(cat(seq, Gate*cross))`),
 * as two HBM-bound kernels instead of 128x128-tile GEMMs on a 13-wide output:
 *   fwd: logits f32 [M,C] = [seq | gated] . W^T + bias      (seq, gated bf16 [M,H] contiguous; W bf16 [C,2H])
 *   bwd: dl bf16 [M, ldd >= C] -> dseq bf16 [M,H] (gradient w.r.t. seq through the classifier) and, for the gated
 *        half, the gate backward applied in place: du = dgated*cross*g*(1-g), dcross = dgated*g (as icka_gate_bwd);
 *        dW / db leave as icka_cls_head_bwd_slabs(M) slabs of icka_cls_head_slab_floats(H,C) floats each
 *        ([C*2H] weight part, then [16] bias part), to be summed by icka_gemm_grouped_ex slab reductions.
 * Needs 2H/8 <= 256 (H <= 1024). */
int icka_cls_head_fwd(const void* seq, const void* gated, const void* W, const float* bias, float* logits, int32_t M,
                      int32_t H, int32_t C, void* stream);
/* "mixed16" form of icka_cls_head_fwd: seq, gated and W are IEEE fp16 (the fp16 twins of the encoder outputs, the fp16
 * gate product of icka_gemm and the fp16 weight shadow). */
int icka_cls_head_fwd_h(const void* seq, const void* gated, const void* W, const float* bias, float* logits, int32_t M,
                        int32_t H, int32_t C, void* stream);
int icka_cls_head_bwd(const void* dl, int64_t ldd, const void* seq, const void* gated, const void* gate,
                      const void* cross, const void* W, void* dseq, void* du, void* dcross, float* partials, int32_t M,
                      int32_t H, int32_t C, void* stream);
int32_t icka_cls_head_bwd_slabs(int32_t M);
int64_t icka_cls_head_slab_floats(int32_t H, int32_t C);
/* Per-sample gates (fusion.hip).  a / c / out are token-major bf16 [B*S, H] with row strides.
 *  mode 0 (Cross_Modal_Interaction_Module.py:1029-1036): g_b = sigmoid(gate[b]);  out = g_b*a + (1-g_b)*c
 *  mode 1 (gate_cl_modeling.py:1369-1373): g_b = softmax(gate[b,0..1])[1];        out = g_b*a   (c = NULL)
 * Backward: da = g*dout, dc = (1-g)*dout (mode 0), and dgate (f32 [B] or [B,2], ZEROED BY THE CALLER) += its gradient. */
int icka_sample_gate_fwd(const void* a, int64_t lda, const void* c, int64_t ldc, const float* gate, int32_t mode,
                         void* out, int64_t ldo, int32_t B, int32_t S, int32_t H, void* stream);
/* "mixed16" forward of a per-sample gate without blend operand (mode 1 = gate_cl_modeling.py:1369-1373, cross = P * cross):
 * a16 fp16 [B*S, H] (row stride lda) -> out bf16 and out16 fp16 (row stride ldo), both rounded from the f32 product. */
int icka_sample_gate_fwd_h(const void* a16, int64_t lda, const float* gate, int32_t mode, void* out, void* out16, int64_t ldo,
                           int32_t B, int32_t S, int32_t H, void* stream);
int icka_sample_gate_bwd(const void* dout, int64_t lddo, const void* a, int64_t lda, const void* c, int64_t ldc,
                         const float* gate, int32_t mode, void* da, int64_t ldda, void* dc, int64_t lddc, float* dgate,
                         int32_t B, int32_t S, int32_t H, void* stream);
/* crs_classifier of gate_cl (gate_cl_modeling.py:1258,:1364-1366): crs[b,:] (f32 [B,2], ZEROED BY THE CALLER) +=
 * W[2, S*2H] . cat(seq,cross)[b].view(-1) + bias.  W bf16 (shadow), bias f32[2].  Backward: dseq / dcross (bf16
 * contiguous [B*S,H]) are the input gradients, dW f32 [2, S*2H] and dbias f32[2] follow `accumulate`. */
int icka_crs_fwd(const void* seq, int64_t lds, const void* cross, int64_t ldc, const void* W, const float* bias,
                 float* crs, int32_t B, int32_t S, int32_t H, void* stream);
int icka_crs_bwd(const float* dcrs, const void* seq, int64_t lds, const void* cross, int64_t ldc, const void* W,
                 void* dseq, void* dcross, float* dW, float* dbias, int32_t B, int32_t S, int32_t H,
                 int32_t accumulate, void* stream);
/* c = a + b (bf16, contiguous n elements; gradient fan-in). */
int icka_add_bf16(const void* a, const void* b, void* c, int64_t n, void* stream);
/* dz = dg * gelu'(z), contiguous bf16 (16-byte aligned): backward of BertIntermediate's erf-GELU (:548-551) when the
 * sub-module is called on its own; inside a layer it is the ICKA_EPI_DGELU epilogue. */
int icka_dgelu_bf16(const void* dg, const void* z, void* dz, int64_t n, void* stream);
/* dx = dy * (1 - y*y) (bf16, contiguous n elements): backward of the Tanh inside the prompt mapping networks
 * (Cross_Modal_Interaction_Module.py:914-928: Dropout, Linear, Tanh, Dropout, Linear). */
int icka_tanh_bwd(const void* dy, const void* y, void* dx, int64_t n, void* stream);
/* Token-level cross-entropy over valid tokens (benchmark loss), fused forward + backward:
 * logits f32 [M,C] (ld), labels/mask int64 [M]; loss_sum f32[1] += sum of -log p ; count f32[1] += #valid ;
 * dlogits bf16 [M, ldd] (ldd >= C, pad columns zeroed) = (softmax - onehot) * valid  (see icka_scale_by_ratio). */
int icka_token_ce(const float* logits, int64_t ld, const int64_t* labels, const int64_t* mask, float* loss_sum,
                  float* count, void* dlogits, int64_t ldd, int32_t M, int32_t C, void* stream);
/* y = x * num[0] / max(den[0], 1) for bf16 [n] with DEVICE scalars (NULL = 1): applies dloss / #valid to the
 * logit gradients without a host sync.  y may alias x. */
int icka_scale_by_ratio(const void* x, void* y, const float* num, const float* den, int64_t n, void* stream);
/* The same loss without pre-zeroed accumulators and without atomics (fixed summation order, bitwise reproducible):
 * stats f32[3] = {sum of token losses, number of valid tokens, mean loss}; dlogits [M, ldd] unscaled, bf16 or f32
 * (dl_is_f32: autograd wants the gradient of the f32 logits in f32); partials = f32 workspace of
 * icka_token_ce_workspace_floats(M).  Two launches (per-block partials, finalize). */
int64_t icka_token_ce_workspace_floats(int32_t M);
int icka_token_ce_fused(const float* logits, int64_t ld, const int64_t* labels, const int64_t* mask, float* stats,
                        float* partials, void* dlogits, int64_t ldd, int32_t dl_is_f32, int32_t M, int32_t C, void* stream);
/* p[0..n) = 0 (16-byte aligned p): clears the atomically accumulated embedding-table gradients. */
int icka_zero_f32(float* p, int64_t n, void* stream);
/* out[0] = num[0] / max(den[0], 1)   (mean loss from the two accumulators of icka_token_ce). */
int icka_scalar_ratio(float* out, const float* num, const float* den, void* stream);
/* ---------------------------------------------------------------------------------------------------------------
 * Linear-chain CRF over the per-token emissions.  The reference calls the third-party
 * `torchcrf.CRF(num_tags, batch_first=True)` (Cross_Modal_Interaction_Module.py:911, :1045-1057;
 * my_bert/cl_modeling.py:1269, :1380-1386); these entry points follow that package's algorithm (pytorch-crf 0.7.2).
 * emissions f32 [B,S,C] (row stride ld >= C), tags / mask int64 [B,S] (mask NULL = all on; step 0 is always on),
 * start/end f32 [C], trans f32 [C,C] (from, to).  C <= 64; one wave per sample.
 *   icka_crf_llh   : llh[b] = score(gold path) - log Z
 *   icka_crf_grad  : d_emissions[b,t,:] = gllh[b] * d llh / d e  (written), d_start / d_end / d_trans += (atomics)
 *   icka_crf_decode: Viterbi path per sample, best_tags [B,S] (-1 past sum(mask)), best_score [B] (nullable)
 * icka_crf_grad and icka_crf_decode keep S*C values per sample in LDS: S*C <= 12288. */
int icka_crf_llh(const float* emissions, int64_t ld, const int64_t* tags, const int64_t* mask, const float* start,
                 const float* end, const float* trans, float* llh, int32_t B, int32_t S, int32_t C, void* stream);
int icka_crf_grad(const float* emissions, int64_t ld, const int64_t* tags, const int64_t* mask, const float* start,
                  const float* end, const float* trans, const float* gllh, float* d_emissions, int64_t ldd,
                  float* d_start, float* d_end, float* d_trans, int32_t B, int32_t S, int32_t C, void* stream);
int icka_crf_decode(const float* emissions, int64_t ld, const int64_t* mask, const float* start, const float* end,
                    const float* trans, int64_t* best_tags, float* best_score, int32_t B, int32_t S, int32_t C,
                    void* stream);
/* ---------------------------------------------------------------------------------------------------------------
 * Bidirectional single-layer LSTM of the tagging tail:
 * nn.LSTM(H, H, batch_first=True, bidirectional=True) at Cross_Modal_Interaction_Module.py:905-908, called :1042
 * (`x, _ = self.lstm(result)`).  Gate order i, f, g, o as torch.nn.LSTM.  The input projection (x . W_ih^T + b_ih +
 * b_hh for both directions, f32 [B*S, 8H], column = dir*4H + gate*H + unit) is an icka_gemm; these entry points run
 * the recurrence, one launch per time step (both directions per launch), B <= 64, H % 32 == 0.
 *   icka_lstm_fwd: w_hh bf16 [2][4H][H]; writes y bf16 [B,S,2H] (h_t, the LSTM output), c_all f32 [B,S,2,H],
 *                  act bf16 [B,S,2,4H] (gate activations) and optionally hprev bf16 [B,S,2H] (h_{t-1} per step: the
 *                  operand of the dW_hh GEMM).
 *   icka_lstm_bwd: dy bf16 [B,S,2H], w_hh_t = W_hh^T bf16 [2][H][4H] (icka_transpose_bf16); writes the gate
 *                  pre-activation gradients dgates bf16 [B*S, ldg >= 8H]; dW_ih, dW_hh, db and dx are GEMMs / column
 *                  sums over dgates afterwards.  dc_carry f32 [2,B,H] is workspace. */
int icka_lstm_fwd(const float* gates_x, int64_t ldg, const void* w_hh, void* y, float* c_all, void* act, void* hprev,
                  int32_t B, int32_t S, int32_t H, int32_t flags, void* stream);
int icka_lstm_bwd(const void* dy, const void* w_hh_t, const void* act, const float* c_all, void* dgates, int64_t ldg,
                  float* dc_carry, int32_t B, int32_t S, int32_t H, int32_t flags, void* stream);
/* flags of icka_lstm_fwd / icka_lstm_bwd (0 = the defaults; per call, no process-wide switch):
 *   default            for B <= 32 and H <= 768 the recurrence is ONE persistent launch per direction pair -- W_hh slices and cell
 *                      state resident in registers; the blocks hand h_t over as flag-in-data 8-byte words polled by the consumers
 *                      (H % 256 == 0 shapes; others use the ticket form): forward broadcasts h_t, backward reduce-scatters bf16
 *                      partial products of dgates_t . W_hh; in the forward a batch of 17..32 rows runs as two independent tiles
 *                      of 16 rows (separate blocks).
 *   ICKA_LSTM_PER_STEP        one launch per time step instead of the persistent launch.
 *   ICKA_LSTM_TICKETS         persistent launch with step tickets + L1 invalidate instead of tagged words (same forward
 *                             results; backward gradients agree to bf16 rounding).
 *   ICKA_LSTM_NO_BATCH_SPLIT  one block per 16 hidden units holds all rows (same results).
 * icka_lstm_barrier_error() returns 1 if a hand-off wait ever gave up (bounded spin: the kernel then finishes with NaN-poisoned
 * data instead of hanging), -1 if the query itself failed. */
#define ICKA_LSTM_PER_STEP 1
#define ICKA_LSTM_TICKETS 2
#define ICKA_LSTM_NO_BATCH_SPLIT 4
/* The persistent launches wait on words written by other blocks of the same grid, with a bounded spin.  A wait that gives
 * up (a peer block that never became resident, a lost word) raises a sticky error word AND poisons the recurrence: the
 * waiting block continues with NaN operands, so every later step of every block -- y, the loss, every gradient of the
 * call -- is NaN: a failed launch never passes for a result.  icka_lstm_barrier_error() returns 1 once that has happened,
 * 0 otherwise, -1 if the query itself failed; the word lives in host memory mapped into the device, so the query costs no
 * device synchronisation and the host side (icka_amd/lstm.py, icka_amd/graph.py) polls it at every touch-point and raises.
 * icka_lstm_clear_error() resets it.  (The reference's nn.LSTM, Cross_Modal_Interaction_Module.py:905-908 / :1042, cannot
 * fail this way; this is the price of the one-launch recurrence.) */
int icka_lstm_barrier_error(void);
int icka_lstm_clear_error(void);
/* CUs to keep free of persistent LSTM blocks (default 0): the persistent forms are used only when their grid fits into the
 * device's CUs minus this reserve, else the one-launch-per-step form runs.  dp.GradReducer reserves the CUs RCCL's
 * all-reduce workgroups may occupy on the communication stream while the recurrence runs. */
int icka_lstm_set_reserved_cus(int32_t n);
/* Test hooks of the give-up path (tests/test_lstm_gpu.py): poll_limit > 0 replaces the poll budgets of all persistent
 * forms (0 restores the defaults); drop_step >= 0 makes block (0, direction 0, batch tile 0) skip publishing that step so
 * that its consumers time out (-1: off). */
int icka_lstm_test_hooks(int32_t poll_limit, int32_t drop_step);
/* y bf16 [M <= 64, N] = act(x . W^T + bias) for a handful of rows (BertPooler.forward :675-681: tanh(dense(h[:, 0])),
 * 32 rows at c2): x bf16 rows with stride ldx, W bf16 [N,K], K % 128 == 0; act 0 = none, 1 = tanh. */
int icka_linear_small_m(const void* x, int64_t ldx, const void* W, const float* bias, void* y, int64_t ldy, int32_t M,
                        int32_t N, int32_t K, int32_t act, void* stream);
/* out[b][c][r] = in[b][r][c] for bf16 matrices (batch of [R,C]). */
int icka_transpose_bf16(const void* in, void* out, int32_t batch, int32_t R, int32_t C, void* stream);
/* ---------------------------------------------------------------------------------------------------------------
 * Frozen ResNet image encoder, forward only (resnet/resnet.py:57-150 Bottleneck / ResNet,
 * resnet/resnet_utils.py:13-53 myResnet.forward).  Convolutions run as icka_gemm calls on NHWC bf16 activations with
 * eval-mode BatchNorm folded into weight / bias and ReLU / residual add in the epilogue (ICKA_EPI_RELU,
 * ICKA_EPI_ADD_RELU); these entry points build the operands.  rows_padded >= B*Ho*Wo: rows past the end are zeroed so
 * the GEMM M dimension can be a multiple of 128.
 *   stem_patches : image f32 NCHW [B,3,H,W] -> 7x7/s2/p3 patches bf16 [rows_padded, 192], k = (ky*7+kx)*3 + c, 147..191 = 0
 *   im2col3x3    : NHWC bf16 [B,H,W,C] -> [rows_padded, 9*C], k = (ky*3+kx)*C + c, pad 1, stride 1 or 2
 *   subsample    : rows (stride*y, stride*x) of an NHWC map (strided 1x1 convolution input)
 *   maxpool3x3s2 : nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
 *   features_out : x bf16 [B,P,C] -> att f32 [B,C,P] (NCHW), fc f32 [B,C] = mean over P, tokens bf16 [B*P,C] (nullable) */
int icka_conv_stem_patches(const float* image, void* patches, int32_t B, int32_t H, int32_t W, int64_t rows_padded,
                           void* stream);
int icka_conv_im2col3x3(const void* src, void* patches, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                        int64_t rows_padded, void* stream);
/* The 3x3 / pad 1 convolution itself (stride 1 or 2) as an IMPLICIT GEMM: y bf16 [rows_padded, Cout] = epilogue(patches(x) .
 * w^T + bias (+ aux)), x NHWC bf16 [B,H,W,C] (C % 64 == 0), w bf16 [Cout, 9*C] with k = (ky*3+kx)*C + c (Cout % 64 == 0),
 * epilogue ICKA_EPI_NONE / RELU / ADD / ADD_RELU (aux bf16 [rows_padded, ldaux]), zeros = at least 128 B of zero bytes.
 * Same result as icka_conv_im2col3x3 + icka_gemm without the 9x patch matrix (resnet/resnet.py:63-64 conv2 + bn2 + relu). */
int icka_conv3x3_gemm(const void* x, const void* w, const float* bias, const void* aux, int64_t ldaux, void* y, int32_t B,
                      int32_t H, int32_t W, int32_t C, int32_t Cout, int32_t stride, int64_t rows_padded, int32_t epilogue,
                      const void* zeros, void* stream);
int icka_conv_subsample(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, int32_t stride,
                        int64_t rows_padded, void* stream);
int icka_conv_maxpool3x3s2(const void* src, void* dst, int32_t B, int32_t H, int32_t W, int32_t C, int64_t rows_padded,
                           void* stream);
int icka_conv_features_out(const void* x, float* att, float* fc, void* tokens, int32_t B, int32_t P, int32_t C,
                           void* stream);
/* Dropout nonce for hipGraph replay.  Every dropout-bearing kernel XORs two DEVICE words into its (by-value) seed at
 * entry when a nonce is registered.  A captured graph re-launches the same seed values, so the graph also captures
 * icka_bump_dropout_nonce at the start of a step: each replay then draws fresh masks, and the forward and backward
 * kernels of one replay still see the same nonce.  NULL (default) disables it.  Process-global. */
int icka_set_dropout_nonce(const uint32_t* device_words);
int icka_bump_dropout_nonce(uint32_t* device_words, void* stream);
/* Debug/test helper: materialise the dropout keep-multiplier (0 or 1/(1-p)) for element indices [0,n) as f32. */
int icka_dropout_mask(float* out, int64_t n, float p_drop, uint64_t seed, void* stream);
/* The same for the attention-probability dropout of icka_attn_fwd / icka_attn_bwd (and the fp32-mode softmax): the multiplier
 * of element (row, key) of a [rows, Skv] probability matrix, rows = (batch * heads + head) * Sq + query.  These sites draw two
 * decisions from one 32-bit hash -- hash(row * Skv + key / 2), low 16 bits for even keys, high 16 bits for odd keys -- so
 * their mask is NOT icka_dropout_mask of the flat index. */
int icka_attn_dropout_mask(float* out, int64_t rows, int32_t Skv, float p_drop, uint64_t seed, void* stream);

/* ===============================================================================================================
 * fp32 "exact" mode (icka_amd.set_precision(model, "fp32")): the same path in f32 storage and f32 arithmetic, for
 * BASELINE.json's "within 1e-3 fp32" bar -- the reference is fp32 end to end (SURVEY.md section 0;
 * Cross_Modal_Interaction_Module.py:950).  Every contraction runs on the f32-input matrix instruction
 * v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain); attention materialises its scores as the reference does
 * (:488-502).  All pointers are f32 device pointers unless stated.  Used only when selected; never a fallback.
 *
 * Batched GEMM  C[b0,b1] = alpha * op(A[b0,b1]) . op(B[b0,b1]) (+ bias[n]) (+ beta * C[b0,b1]),  nb0*nb1 <= 65535:
 *   NT: A[M,K] B[N,K]    NN: A[M,K] B[K,N]    TN: A[K,M] B[K,N]    TT: A[K,M] B[N,K]
 * any M, N, K >= 1, any leading dimension; X[b0,b1] = X + b0*x_bs0 + b1*x_bs1 (elements).  Replaces nn.Linear /
 * torch.matmul (:479-481, :488, :502, :533, :549, :562, :958; cl_modeling.py:1363, :1371) and their autograd. */
enum { ICKA_GEMM_TT = 3 };
typedef struct icka_xgemm_desc {
    int32_t op, M, N, K;
    const float* A; int64_t lda, a_bs0, a_bs1;
    const float* B; int64_t ldb, b_bs0, b_bs1;
    float* C; int64_t ldc, c_bs0, c_bs1;
    int32_t nb0, nb1;
    const float* bias;      /* f32 [N] or NULL */
    float alpha, beta;
} icka_xgemm_desc;
int icka_x_gemm(const icka_xgemm_desc* d, void* stream);
/* y = LayerNorm(dropout(x) + residual) * gamma + beta  (BertSelfOutput :561-565, BertOutput :532-536, BertLayerNorm
 * :518-522; x is the dense output WITH its bias).  xhat [M,H] / rstd [M] saved for backward (nullable). */
int icka_x_ln_fwd(const float* x, int64_t ldx, const float* residual, int64_t ldr, const float* gamma,
                  const float* beta, float* y, float* xhat, float* rstd, int32_t M, int32_t H, float eps,
                  float p_drop, uint64_t seed, void* stream);
/* dpre [M,H] = gradient of the LayerNorm input (= of the residual); ddense (nullable) = dpre * dropout mask. */
int icka_x_ln_bwd(const float* dy, int64_t lddy, const float* xhat, const float* rstd, const float* gamma,
                  float* dpre, float* ddense, int32_t M, int32_t H, float p_drop, uint64_t seed, void* stream);
/* out[c] (+)= sum_r a[r,c] * (b ? b[r,c] : 1)   (bias and LayerNorm parameter gradients; fixed summation order) */
int icka_x_colsum(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int32_t M, int32_t N,
                  int32_t accumulate, void* stream);
/* BertEmbeddings.forward :398-410 up to the LayerNorm (dropout: icka_x_dropout); scatter = backward into the tables
 * from the LayerNorm-input gradient (f32 atomics, padding_idx row of the word table skipped). */
int icka_x_embed_fwd(const int64_t* ids, const int64_t* token_type, const float* word, const float* pos,
                     const float* typ, const float* gamma, const float* beta, float* y, float* xhat, float* rstd,
                     int32_t B, int32_t S, int32_t H, float eps, void* stream);
int icka_x_embed_scatter(const float* dpre, const int64_t* ids, const int64_t* token_type, float* dword, float* dpos,
                         float* dtyp, int32_t B, int32_t S, int32_t H, int32_t padding_idx, void* stream);
/* In place P[B,heads,Sq,Skv] = softmax(P*scale + add_mask[b,j]) (:489-494); Pd = dropout(P) (:500) when p_drop > 0.
 * Backward, in place on dS (which holds dPd on entry): dS = P * (dPd*mask - sum_j dPd*mask*P) * scale. */
int icka_x_softmax_fwd(float* P, float* Pd, const float* add_mask, int32_t B, int32_t heads, int32_t Sq, int32_t Skv,
                       float scale, float p_drop, uint64_t seed, void* stream);
int icka_x_softmax_bwd(const float* P, float* dS, int32_t B, int32_t heads, int32_t Sq, int32_t Skv, float scale,
                       float p_drop, uint64_t seed, void* stream);
/* Elementwise activations.  mode 0: y = gelu(x) (erf form, :31-37); 1: y = tanh(x) (BertPooler :675-681);
 * 2: y2 = sigmoid(x), y = y2 * aux (cl_modeling.py:1363-1367).
 * Backward: mode 0: dx = dy * gelu'(saved = x); 1: dx = dy * (1 - saved^2), saved = y; 2 (saved = gate, aux = cross):
 * dx = dy * aux * g(1-g) (gradient of the gate pre-activation), dx2 = dy * g (gradient of cross through the product). */
int icka_x_act_fwd(const float* x, const float* aux, float* y, float* y2, int64_t n, int32_t mode, void* stream);
int icka_x_act_bwd(const float* dy, const float* saved, const float* aux, float* dx, float* dx2, int64_t n,
                   int32_t mode, void* stream);
/* y = x * keep(i) / (1-p), the same counter-hash mask as the bf16 kernels (forward and backward are the same call) */
int icka_x_dropout(const float* x, float* y, int64_t n, float p_drop, uint64_t seed, void* stream);
int icka_x_add(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int32_t M,
               int32_t N, void* stream);
/* out [M, Ha+Hb] = [a | b]  (torch.cat((seq, cross), dim=-1), cl_modeling.py:1363-1370) */
int icka_x_concat2(const float* a, int64_t lda, int32_t Ha, const float* b, int64_t ldb, int32_t Hb, float* out,
                   int32_t M, void* stream);
/* f32 twin of icka_regions_to_tokens */
int icka_x_regions_to_tokens(const float* src, float* dst, int32_t B, int32_t R, int32_t C, int32_t layout,
                             void* stream);
/* f32 twins of icka_sample_gate_fwd / _bwd (dgate is OVERWRITTEN: one block per sample, fixed order) */
int icka_x_sample_gate_fwd(const float* a, int64_t lda, const float* c, int64_t ldc, const float* gate, int32_t mode,
                           float* out, int64_t ldo, int32_t B, int32_t S, int32_t H, void* stream);
int icka_x_sample_gate_bwd(const float* dout, int64_t lddo, const float* a, int64_t lda, const float* c, int64_t ldc,
                           const float* gate, int32_t mode, float* da, int64_t ldda, float* dc, int64_t lddc,
                           float* dgate, int32_t B, int32_t S, int32_t H, void* stream);
/* BiLSTM recurrence of the fp32 mode (nn.LSTM(H, H, batch_first=True, bidirectional=True), :905-908, call :1042).  The
 * gate pre-activations of step k are accumulated into gates [B*S, 8H] (= x.W_ih^T + b_ih + b_hh for all steps, both
 * directions: [i f g o | i f g o]) by icka_x_gemm with beta = 1 (h_{t-1} . W_hh^T, batched over the two directions);
 * cell_fwd then applies the cell update of step k for both directions (forward direction t = k, reverse t = S-1-k):
 * gates <- activations (saved), c_all / y [B*S, 2H] <- c_t / h_t, hprev <- the h_{t-1} it used (zeros at k = 0).
 * cell_bwd (k = S-1 .. 0): act <- gate pre-activation gradients in place, from dy, the gradient dh_rec [2,B,H] that
 * reached h_t through the next step's recurrent product (first != 0: none yet) and the cell-state carry dc_carry [2,B,H]. */
int icka_x_lstm_cell_fwd(float* gates, float* c_all, float* y, float* hprev, int32_t B, int32_t S, int32_t H, int32_t k,
                         void* stream);
int icka_x_lstm_cell_bwd(const float* dy, const float* dh_rec, float* dc_carry, float* act, const float* c_all, int32_t B,
                         int32_t S, int32_t H, int32_t k, int32_t first, void* stream);
/* f32 twins of icka_embed_prompt_fwd / _bwd (the LayerNorm parameter gradients come from icka_x_colsum; the scatter
 * takes the LayerNorm-input gradient; dprompt f32 [B,P,H], every row written exactly once) */
int icka_x_embed_prompt_fwd(const int64_t* ids, const int32_t* src, const float* prompt, const float* word, const float* pos,
                            const float* typ, const float* gamma, const float* beta, float* y, float* xhat, float* rstd,
                            int32_t B, int32_t S_in, int32_t S, int32_t P, int32_t H, int32_t pos_offset, float eps,
                            void* stream);
int icka_x_embed_prompt_scatter(const float* dpre, const int64_t* ids, const int32_t* src, float* dword, float* dpos,
                                float* dtyp, float* dprompt, int32_t B, int32_t S_in, int32_t S, int32_t P, int32_t H,
                                int32_t pos_offset, int32_t padding_idx, void* stream);
/* f32 twins of icka_token_ce / icka_scale_by_ratio (dlogits f32 [M,C] contiguous, unscaled) */
int icka_x_token_ce(const float* logits, int64_t ld, const int64_t* labels, const int64_t* mask, float* loss_sum,
                    float* count, float* dlogits, int32_t M, int32_t C, void* stream);
int icka_x_scale_by_ratio(const float* x, float* y, const float* num, const float* den, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Data-parallel helpers (csrc/dp.hip).  Reference: apex DistributedDataParallel all-reduces the gradients of every
 * parameter during backward (My_cross_attention.py:768-776, process group :653-657, launch line :1104); here the exchange
 * is RCCL over xGMI driven by icka_amd/dp.py, and these entry points are what runs around it on the GPU.
 *
 * Wire format: gradients travel as bf16.  Matrix gradients get their wire copy from the weight-gradient GEMM itself
 * (icka_gemm_desc.C3 beside an f32 C: the value after beta-accumulation); everything else in a bucket is cast by ONE launch
 * over a chunk table:
 *   icka_dp_cast_chunks: table_dev = n_chunks x {first element, element count} (int64 pairs in device memory; counts are
 *     multiples of 8 and at most icka_dp_chunk_elems()); dst[i] = bf16(src[i]) for every i in a chunk; src / dst are the
 *     BASES of the flat f32 gradient buffer / the bf16 wire buffer (same element offsets).
 *   icka_dp_cast_back_scaled: dst f32 [n] = src bf16 [n] * scale (the reduced bucket back into the gradient buffer, with the
 *     1 / world factor of the mean folded in: the all-reduce is a SUM).
 * Bucket-ready flags for a step captured as ONE hipGraph with eager collectives (icka_amd/graph.py: FlaggedStep):
 *   icka_dp_step_bump(step_word): step_word[0] += 1 -- first node of the graph;
 *   icka_dp_flag_set(flag_word, step_word): flag_word[0] = step_word[0] -- a node right after the kernel that finishes the
 *     bucket's last gradient;
 *   icka_dp_flag_wait(flag_word, tag, bad_word, max_polls): a one-wave kernel for the COMMUNICATION stream that returns once
 *     flag_word[0] >= tag (wrap-safe), polling with s_sleep at most max_polls times; if it gives up it raises the error word
 *     read by icka_dp_error() (host memory mapped into the device: no synchronisation) and, when bad_word (device memory,
 *     one uint32 per bucket) is given, stores tag there.  The NaN itself is written by kernels ordered AFTER everything
 *     that could overwrite it:
 *   icka_dp_poison_if(bad_word, tag, target, target_is_bf16, n): if bad_word[0] == tag, NaN into target[0 .. n) (n <= 64;
 *     bf16 or f32 elements) -- on the communication stream between the bucket's chunk cast and its all-reduce (the NaN
 *     reaches every rank through the sum) and again after the cast-back;
 *   icka_dp_poison_final(bad_words, n_buckets, tag, gflat, starts_dev, n): one launch on the COMPUTE stream after the join
 *     that ends the step (the graph's late gradient stores are over): for every bucket b with bad_words[b] == tag, NaN into
 *     gflat[starts_dev[b] .. + n).
 *   icka_dp_init() maps the error word (not inside a stream capture); icka_dp_clear_error() resets it. */
int64_t icka_dp_chunk_elems(void);
int icka_dp_cast_chunks(const float* src, void* dst_bf16, const int64_t* table_dev, int32_t n_chunks, void* stream);
int icka_dp_cast_back_scaled(const void* src_bf16, float* dst, int64_t n, float scale, void* stream);
int icka_dp_init(void);
int icka_dp_error(void);
int icka_dp_clear_error(void);
int icka_dp_step_bump(void* step_word, void* stream);
int icka_dp_flag_set(void* flag_word, const void* step_word, void* stream);
int icka_dp_flag_wait(const void* flag_word, uint32_t tag, void* bad_word, int32_t max_polls, void* stream);
int icka_dp_poison_if(const void* bad_word, uint32_t tag, void* target, int32_t target_is_bf16, int32_t n, void* stream);
int icka_dp_poison_final(const void* bad_words, int32_t n_buckets, uint32_t tag, float* gflat, const int64_t* starts_dev,
                         int32_t n, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * The parameter update that follows backward in the reference's loop (My_cross_attention.py:831-844: clip_grad_norm_(1.0),
 * AdamW.step() over the two weight-decay groups of :743-751) as three launches over the flat parameter / gradient / state
 * buffers (csrc/optim.hip; icka_amd/optim.py: ArenaAdamW).  Outside the fwd+bwd metric, reported beside it.  Chunk tables as
 * for icka_dp_cast_chunks (int64 pairs {first element, count}, counts multiples of 8, at most icka_optim_chunk_elems()).
 *   icka_optim_sqnorm: partials[b] = sum of squares of grads over chunk b.
 *   icka_optim_clip:   out2[0] = sqrt(sum of the n partials) (fixed order), out2[1] = min(1, max_norm / (norm + 1e-6)) --
 *                      torch.nn.utils.clip_grad_norm_'s coefficient, left on the device (max_norm <= 0: 1).
 *   icka_optim_adamw:  torch.optim.AdamW arithmetic for the chunks of ONE weight-decay group: p *= 1 - lr * wd;
 *                      m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t)
 *                      + eps), with g = grads * clip_coef[0] (clip_coef may be NULL); also writes the bf16 (and fp16) weight
 *                      shadow of every updated element when shadow pointers are given (same element offsets). */
int64_t icka_optim_chunk_elems(void);
int icka_optim_sqnorm(const float* grads, const int64_t* table_dev, int32_t n_chunks, float* partials, void* stream);
int icka_optim_clip(const float* partials, int32_t n, float max_norm, float* out2, void* stream);
int icka_optim_adamw(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, void* shadow_bf16, void* shadow_f16,
                     const int64_t* table_dev, int32_t n_chunks, const float* clip_coef, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ICKA_HIP_H */
