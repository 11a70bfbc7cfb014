"""autograd.Functions of the hot path: one Function per reference block (embeddings, BERT layer, cross-attention
layer, linear, gated head, dropout, pooler, token-CE), each a fixed sequence of libicka_hip.so launches.

Design notes
  * Activations between blocks are bf16 [tokens, hidden] matrices; parameters are NOT autograd inputs: kernels read
    the bf16 shadow of the ParamArena and the backward writes parameter gradients straight into the arena's fp32
    gradient buffer (GEMM beta 0/1), then attaches ``p.grad`` views.  An ``anchor`` leaf keeps autograd engaged.
  * Every dropout site draws a 64-bit seed at forward time and stores it for its backward, which re-generates the
    mask inside the kernels (nothing is materialised).
  * Gradient fan-in (residual + dense paths) is fused into GEMM epilogues (EPI_ADD) or the LN backward (dy2).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import kernels as K
from .arena import ParamArena

BF16, F32, F16 = torch.bfloat16, torch.float32, torch.float16
import os as _os
# The attention forward may leave its dropout keep bits for the backward.  At 128 x 128 the c2 step is 0.4 % slower with them
# (the forward pays for packing and storing the bits, the backward's hashes were hidden behind its MFMA / LDS waits:
# profiles/r03_attn_keepbits_ab.txt); at the 256 x 256 whole-head instance of bert-large the hashes are NOT hidden: backward
# 98.1 -> 87.2 us, forward 43.5 -> 45.4 us, c4 step 23.26 -> 23.17 ms (profiles/r04_attn_keepbits_256.txt).
# "auto" (default): on for self-attention sites with more than 128 keys; "1" / "0": always / never.
ATTN_KEEPBITS = _os.environ.get("ICKA_ATTN_KEEPBITS", "auto")
# dense -> bias + dropout + residual -> LayerNorm as ONE launch where the shape allows it (icka_gemm_ln: one 128-row block per CU;
# bitwise the two launches).  "0" keeps the two launches everywhere (same-box A/B: tools/ab_env.sh).
FUSE_DENSE_LN = _os.environ.get("ICKA_FUSE_DENSE_LN", "1") != "0"
# ... from this many rows on: the fused launch has M / 128 * 8 blocks and its LayerNorm phase runs on exactly those CUs, where the row
# kernel spreads the same rows over the whole chip -- at M = 512 (the reference's test loop at batch 4: 32 blocks) the replayed
# forward is 7 % SLOWER with it (1.25 vs 1.17 ms, profiles/r05_gemm_ln_ab.txt); it pays when the GEMM's grid fills the chip
FUSE_DENSE_LN_MIN_ROWS = 3072
# QKV projection + self-attention as ONE launch where the shape allows it (icka_gemm_qkv_attn: 128 tokens per sample, head size 64;
# bitwise the two launches).  "0" keeps the two launches.  The library itself declines grids below 128 tiles.
FUSE_QKV_ATTN = _os.environ.get("ICKA_FUSE_QKV_ATTN", "1") != "0"
FUSE_QKV_ATTN_MIN_ROWS = 3072


def _keepbits_on(Sq: int, Skv: int) -> bool:
    if ATTN_KEEPBITS == "auto":
        return Sq > 128 and Skv > 128 and Sq <= 256 and Skv <= 256
    return ATTN_KEEPBITS == "1"


def _empty(ref: torch.Tensor, *shape, dtype=BF16) -> torch.Tensor:
    return torch.empty(*shape, dtype=dtype, device=ref.device)


def _c(t: torch.Tensor) -> torch.Tensor:
    """autograd may hand us non-contiguous / broadcast gradients: make them plain row-major bf16."""
    if t.dtype != BF16:
        raise TypeError("expected a bf16 gradient, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


class Dims(object):
    """Static description of one call: batch, query length, kv length, sizes, dropout probabilities."""
    __slots__ = ("B", "S", "R", "H", "I", "heads", "eps", "p_hidden", "p_attn", "train", "h16")

    def __init__(self, B, S, R, H, I, heads, eps, p_hidden, p_attn, train, h16=False):
        self.B, self.S, self.R, self.H, self.I, self.heads, self.eps = B, S, R, H, I, heads, eps
        self.p_hidden = p_hidden if train else 0.0
        self.p_attn = p_attn if train else 0.0
        self.train = train
        # "mixed16": the forward GEMMs of the encoder layers read fp16 operands (activations: the fp16 twin that is also the
        # residual stream; weights: the arena's fp16 shadow); backward stays on the bf16 copies
        self.h16 = bool(h16)


def _fwd_twin(A: ParamArena, x, xf, d: Dims):
    """The forward twin of a block input.  bf16 mode: the f32 residual twin (or None).  mixed16: the fp16 copy that is both
    the forward GEMM operand and the residual (made from the bf16 tensor when the producer did not leave one)."""
    if not d.h16:
        return xf
    if A.shadow16 is None:      # first mixed16 call on this arena: create and fill the fp16 weight shadow
        A.enable_fp16_shadow()
        A.sync(force=True)
    if xf is not None and xf.dtype == F16:
        return xf
    return K.cast_to_f16(x if xf is None else xf, _empty(x, x.shape[0], x.shape[1], dtype=F16))


# Weight gradients (dW = dY^T . X, reduction over the tokens) have no consumer inside backward: each block queues
# them and launches the whole batch as ONE grouped GEMM at the end of its backward (4 GEMMs of 36..144 tiles each
# fill the 256 CUs together instead of one after the other).
def _wgrad(A: ParamArena, dy, x, gout, beta: float, bias=None) -> None:
    """Queue dW (+)= dy^T . x; with ``bias`` (parameter or tuple of adjacent parameters) also db (+)= colsum(dy),
    fused into the same GEMM when the shape is on the fast path, else by the column-sum kernel."""
    bout, bacc = None, False
    if bias is not None:
        bacc = A.grad_beta(bias) > 0
        bout = A.g_cat(bias) if isinstance(bias, tuple) else A.g(bias)
        M, N, Kt = dy.shape[1], x.shape[1], dy.shape[0]
        if M % 128 or N % 128 or Kt % 64:
            csw = A.workspace("colsum", K._lib.load().icka_colsum_workspace_floats(M))
            K.colsum(dy, bout, csw, accumulate=bacc)
            bout = None
    # data parallel with bf16 buckets: the same epilogue writes the bf16 wire copy of the gradient (ParamArena.wire_of)
    wire = A.wire_of(gout, beta)
    A.pending_wgrad.append((K.gemm_desc(K.GEMM_TN, dy, x, gout, beta=beta, colsum_out=bout, colsum_accumulate=bacc,
                                        **wire), dy, x, gout, bout, wire))


def _flush_wgrad(A: ParamArena) -> None:
    """One grouped launch for the block's queued weight gradients; the LayerNorm dgamma / dbeta slab reductions queued
    by _ln_bwd_deferred are summed by extra blocks of the same launch."""
    if A.pending_wgrad or A.pending_reductions:
        K.gemm_grouped([t[0] for t in A.pending_wgrad], reductions=A.pending_reductions)
        A.pending_wgrad = []
        A.pending_reductions = []


def _ln_bwd_deferred(A: ParamArena, tag: str, norm, dy, xhat, rstd, *, dy2, dres, dx, p_drop, seed) -> None:
    """LayerNorm backward of a block: rows now, the parameter-gradient finalize with the block's weight-gradient launch
    (saves one launch per LayerNorm).  ``tag`` keeps the slab workspaces of one block apart."""
    H = dy.shape[1]
    ws = A.workspace(tag, K._lib.load().icka_ln_bwd_workspace_floats(H))
    acc = A.grad_beta((norm.weight, norm.bias)) > 0
    nslab = K.ln_bwd_slabs(dy, xhat, rstd, norm.weight, ws, dy2=dy2, dres=dres, dx=dx, p_drop=p_drop, seed=seed)
    A.pending_reductions.append(K.slab_reduction(ws, nslab, H, (A.g(norm.weight), A.g(norm.bias)), acc))


# =============================================================================================== sub-blocks
# Each reference sub-module is one forward / backward pair of launch sequences; the layer Functions (BertLayerFn,
# CrossLayerFn) chain them with the gradient fan-ins fused into GEMM epilogues, the sub-module Functions further down
# (AttnCoreFn, DenseResidualNormFn, IntermediateFn) expose the same pairs one by one for callers that compose
# BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput themselves (as BertAttention.forward :451-454 does).
def _attn_generic_fwd(qkv, kvbuf, self_attn: bool, add_mask, d: Dims, Skv: int, seed_a: int, ctx, ctx16, save: bool):
    """Attention core for head sizes other than 64: bf16 q / k / v -> f32, scores / softmax (+ dropout) / context on the
    batched f32-input MFMA GEMM and softmax kernels of csrc/exact.hip (exact.py follows :488-505 step by step), context back
    to bf16 (+ fp16 in mixed16).  Returns what the backward needs: ("generic", q32 buffer, kv32 buffer, P, Pd)."""
    import math
    from . import exact as X
    B, h, S, H = d.B, d.heads, d.S, d.H
    dh = H // h
    q32b = K.cast_bf16_to_f32(qkv, torch.empty(qkv.shape, dtype=F32, device=qkv.device))
    if self_attn:
        q, k, v = q32b[:, :H], q32b[:, H:2 * H], q32b[:, 2 * H:]
        kv32b = None
    else:
        kv32b = K.cast_bf16_to_f32(kvbuf, torch.empty(kvbuf.shape, dtype=F32, device=kvbuf.device))
        q, k, v = q32b, kv32b[:, :H], kv32b[:, H:]
    P = torch.empty(B, h, S, Skv, dtype=F32, device=qkv.device)
    X.gemm_raw(X.GEMM_NT, S, Skv, dh, q, q.stride(0), (S * q.stride(0), dh), k, k.stride(0), (Skv * k.stride(0), dh),
               P, Skv, (h * S * Skv, S * Skv), B, h)
    Pd = torch.empty_like(P) if d.p_attn > 0 else None
    X.softmax_fwd(P, Pd, add_mask, B, h, S, Skv, 1.0 / math.sqrt(dh), d.p_attn, seed_a)
    Pu = P if Pd is None else Pd
    ctx32 = torch.empty(B * S, H, dtype=F32, device=qkv.device)
    X.gemm_raw(X.GEMM_NN, S, dh, Skv, Pu, Skv, (h * S * Skv, S * Skv), v, v.stride(0), (Skv * v.stride(0), dh),
               ctx32, H, (S * H, dh), B, h)
    K.cast_f32_to_bf16(ctx32, ctx)
    if ctx16 is not None:
        K.cast_to_f16(ctx32, ctx16)
    return ("generic", q32b, kv32b, P, Pd) if save else None


def _attn_generic_bwd(gen, self_attn: bool, dctx, d: Dims, Skv: int, seed_a: int, dq_out, dkv_out) -> None:
    """Backward of _attn_generic_fwd: dq / dk / dv in f32 (exact.py:_attn_core_bwd's core), written as bf16 into ``dq_out``
    ([M, 3H] fused for self-attention, [M, H] otherwise) and ``dkv_out`` ([rows, 2H], co-attention; padded rows stay zero)."""
    import math
    from . import exact as X
    _tag, q32b, kv32b, P, Pd = gen
    B, h, S, H = d.B, d.heads, d.S, d.H
    dh = H // h
    scale = 1.0 / math.sqrt(dh)
    dev = q32b.device
    d32 = K.cast_bf16_to_f32(dctx if dctx.is_contiguous() else dctx.contiguous(), torch.empty(B * S, H, dtype=F32, device=dev))
    if self_attn:
        q, k, v = q32b[:, :H], q32b[:, H:2 * H], q32b[:, 2 * H:]
        g32 = torch.empty(B * S, 3 * H, dtype=F32, device=dev)
        dq, dk, dv = g32[:, :H], g32[:, H:2 * H], g32[:, 2 * H:]
        gkv = None
    else:
        q, k, v = q32b, kv32b[:, :H], kv32b[:, H:]
        g32 = torch.empty(B * S, H, dtype=F32, device=dev)
        gkv = torch.empty(kv32b.shape, dtype=F32, device=dev)
        K.zero_(gkv.view(-1))
        dq, dk, dv = g32, gkv[:, :H], gkv[:, H:]
    pb = (h * S * Skv, S * Skv)
    Pu = P if Pd is None else Pd
    dS = torch.empty_like(P)
    X.gemm_raw(X.GEMM_NT, S, Skv, dh, d32, H, (S * H, dh), v, v.stride(0), (Skv * v.stride(0), dh), dS, Skv, pb, B, h)
    X.gemm_raw(X.GEMM_TN, Skv, dh, S, Pu, Skv, pb, d32, H, (S * H, dh), dv, dv.stride(0), (Skv * dv.stride(0), dh), B, h)
    X.softmax_bwd(P, dS, B, h, S, Skv, scale, d.p_attn, seed_a)
    X.gemm_raw(X.GEMM_NN, S, dh, Skv, dS, Skv, pb, k, k.stride(0), (Skv * k.stride(0), dh), dq, dq.stride(0),
               (S * dq.stride(0), dh), B, h)
    X.gemm_raw(X.GEMM_TN, Skv, dh, S, dS, Skv, pb, q, q.stride(0), (S * q.stride(0), dh), dk, dk.stride(0),
               (Skv * dk.stride(0), dh), B, h)
    K.cast_f32_to_bf16(g32, dq_out)
    if gkv is not None:
        K.cast_f32_to_bf16(gkv, dkv_out)


def _attn_core_fwd(A: ParamArena, sa, x, kv_src, add_mask, d: Dims, Skv: int, save: bool, x16=None, kv16=None):
    """BertSelfAttention / BertCoAttention (:478-506, :590-624): fused projection GEMM -> fused attention kernel.
    Returns (ctx bf16 [M,H], ctx fp16 or None, saved).  x16: fp16 copy of x (mixed16) -> the projections of x read fp16
    operands; kv16: fp16 copy of ``kv_src`` -> so does the K/V projection of the co-attention; the projection OUTPUTS
    q / k / v stay bf16 (the attention kernels' operand type)."""
    M, H = x.shape
    h16 = x16 is not None
    xa = x16 if h16 else x
    if kv_src is None:
        qkv = _empty(x, M, 3 * H)
        wq = (sa.query.weight, sa.key.weight, sa.value.weight)
        bq = A.f_cat((sa.query.bias, sa.key.bias, sa.value.bias))
        pre = None
        want_kb = save and d.p_attn > 0 and _keepbits_on(d.S, Skv)
        if (FUSE_QKV_ATTN and d.S in (128, 256) and Skv == d.S and H == 64 * d.heads and M >= FUSE_QKV_ATTN_MIN_ROWS
                and not (want_kb and d.S != 256)):
            # projection + whole-head attention in ONE launch (icka_gemm_qkv_attn): every 256 x 192 tile = 256 / S samples x one
            # head's q | k | v, the attention runs from the tile's LDS images; bitwise the two launches below
            ctx = _empty(x, M, H)
            ctx16 = _empty(x, M, H, dtype=F16) if h16 else None
            seed_a = A.next_seed() if d.p_attn > 0 else 0
            lse = _empty(x, d.B, d.heads, d.S, dtype=F32) if save else None
            kb = K.attn_keepbits(d.B, d.heads, d.S, Skv, x.device) if want_kb else None
            if K.gemm_qkv_attn(xa, A.w16_cat(wq) if h16 else A.w_cat(wq), bq, qkv, add_mask, ctx, lse, d.B, d.heads, d.S,
                               p_drop=d.p_attn, seed=seed_a, out16=ctx16, keepbits=kb):
                return ctx, ctx16, ((qkv, None, lse, seed_a, kb) if save else None)
            pre = (ctx, seed_a, lse, ctx16, kb)      # not a shape of the fused launch: the two launches below, same seed and buffers
        K.gemm(K.GEMM_NT, xa, A.w16_cat(wq) if h16 else A.w_cat(wq), qkv, bias=bq)
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        kvbuf = None
    else:
        pre = None
        qkv = _empty(x, M, H)
        K.gemm(K.GEMM_NT, xa, A.w16(sa.query.weight) if h16 else A.w(sa.query.weight), qkv, bias=sa.query.bias)
        kvbuf = _empty(x, kv_src.shape[0], 2 * H)
        wkv = (sa.key.weight, sa.value.weight)
        if h16 and kv16 is not None:
            K.gemm(K.GEMM_NT, kv16, A.w16_cat(wkv), kvbuf, bias=A.f_cat((sa.key.bias, sa.value.bias)))
        else:
            K.gemm(K.GEMM_NT, kv_src, A.w_cat(wkv), kvbuf, bias=A.f_cat((sa.key.bias, sa.value.bias)))
        q, k, v = qkv, kvbuf[:, :H], kvbuf[:, H:]
    ctx = pre[0] if pre else _empty(x, M, H)
    ctx16 = pre[3] if pre else (_empty(x, M, H, dtype=F16) if h16 else None)
    seed_a = pre[1] if pre else (A.next_seed() if d.p_attn > 0 else 0)
    if H // d.heads != 64:
        if (kv_src is not None) and bool(getattr(sa, "fp8_scores", False)):
            raise ValueError("the fp8 cross-attention kernels are built for head size 64, got %d" % (H // d.heads))
        # head sizes other than 64 (the reference takes any hidden % heads == 0, :459-462): projections and everything around
        # the attention stay on the 16-bit path, the score / softmax / context core runs on the f32-input MFMA kernels of the
        # fp32 mode (probabilities materialised as the reference does, same dropout hash)
        gen = _attn_generic_fwd(qkv, kvbuf, kv_src is None, add_mask, d, Skv, seed_a, ctx, ctx16, save)
        return ctx, ctx16, ((qkv, kvbuf, gen, seed_a, None) if save else None)
    lse = pre[2] if pre else (_empty(x, d.B, d.heads, d.S, dtype=F32) if save else None)
    # BASELINE config c5: a co-attention module flagged fp8_scores runs QK^T / PV on the fp8 matrix cores
    fp8 = (kv_src is not None) and bool(getattr(sa, "fp8_scores", False))
    # optional (ATTN_KEEPBITS): the forward leaves the keep decisions of its probability dropout as bits and the backward reads
    # them instead of hashing every (query, key) element a second time
    kb = pre[4] if pre else (K.attn_keepbits(d.B, d.heads, d.S, Skv, x.device) if (save and d.p_attn > 0 and _keepbits_on(d.S, Skv)) else None)
    K.attn_fwd(q, k, v, add_mask, ctx, lse, d.B, d.heads, d.S, Skv, p_drop=d.p_attn, seed=seed_a, fp8=fp8, out16=ctx16,
               keepbits=kb)
    return ctx, ctx16, ((qkv, kvbuf, lse, seed_a, kb) if save else None)


def _attn_core_bwd(A: ParamArena, sa, x, kv_src, add_mask, d: Dims, Skv: int, saved, ctx, dctx, dres, need_dkv_src: bool):
    """Returns (dx, dkv_src); ``dres`` (optional) is added to dx in the last GEMM's epilogue (residual fan-in)."""
    qkv, kvbuf, lse, seed_a, kb = saved
    M, H = x.shape
    generic = isinstance(lse, tuple)       # head size != 64: the saved f32 operands / probabilities of _attn_generic_fwd
    delta = None if generic else _empty(x, d.B, d.heads, d.S, dtype=F32)
    epi = dict(epilogue=K.EPI_ADD, aux=dres) if dres is not None else {}
    if kv_src is None:
        dqkv = _empty(x, M, 3 * H)
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        if generic:
            _attn_generic_bwd(lse, True, dctx, d, Skv, seed_a, dqkv, None)
        else:
            K.attn_bwd(q, k, v, add_mask, ctx, dctx, lse, delta, dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:], d.B,
                       d.heads, d.S, Skv, p_drop=d.p_attn, seed=seed_a, keepbits=kb)
        wg = (sa.query.weight, sa.key.weight, sa.value.weight)
        bg = (sa.query.bias, sa.key.bias, sa.value.bias)
        _wgrad(A, dqkv, x, A.g_cat(wg), A.grad_beta(wg), bias=bg)
        dx = _empty(x, M, H)
        K.gemm(K.GEMM_NN, dqkv, A.w_cat(wg), dx, **epi)
        return dx, None
    dq = _empty(x, M, H)
    dkv = _empty(x, kv_src.shape[0], 2 * H)
    if kv_src.shape[0] > d.B * Skv:     # row-padded key/value source (region tokens): the attention writes real rows only
        K.zero_rows_(dkv, d.B * Skv)
    if generic:
        _attn_generic_bwd(lse, False, dctx, d, Skv, seed_a, dq, dkv)
    else:
        K.attn_bwd(qkv, kvbuf[:, :H], kvbuf[:, H:], add_mask, ctx, dctx, lse, delta, dq, dkv[:, :H], dkv[:, H:], d.B,
                   d.heads, d.S, Skv, p_drop=d.p_attn, seed=seed_a, keepbits=kb)
    _wgrad(A, dq, x, A.g(sa.query.weight), A.grad_beta(sa.query.weight), bias=sa.query.bias)
    wg = (sa.key.weight, sa.value.weight)
    bg = (sa.key.bias, sa.value.bias)
    _wgrad(A, dkv, kv_src, A.g_cat(wg), A.grad_beta(wg), bias=bg)
    dx = _empty(x, M, H)
    K.gemm(K.GEMM_NN, dq, A.w(sa.query.weight), dx, **epi)
    dsrc = None
    if need_dkv_src:
        dsrc = _empty(x, kv_src.shape[0], H)
        K.gemm(K.GEMM_NN, dkv, A.w_cat(wg), dsrc)
    return dx, dsrc


def _dense_norm_fwd(A: ParamArena, mod, h, res, d: Dims, save: bool, h16=None):
    """BertSelfOutput / BertOutput (:561-565, :532-536): LayerNorm(dropout(dense(h)) + res).  ``res`` is the residual
    input: bf16, its f32 twin, or (mixed16) its fp16 twin.  h16: fp16 copy of h -> the dense GEMM reads fp16 operands and
    the twin of the output is fp16.  Returns (y bf16, y twin, saved)."""
    M = h.shape[0]
    H = mod.dense.weight.shape[0]
    o = _empty(h, M, H, dtype=F32)      # GEMM -> LayerNorm intermediates stay f32 (no extra 16-bit rounding)
    y = _empty(h, M, H)
    yf = _empty(h, M, H, dtype=F16 if d.h16 else F32)
    xhat = _empty(h, M, H) if save else None
    rstd = _empty(h, M, dtype=F32) if save else None
    seed_h = A.next_seed() if d.p_hidden > 0 else 0
    twin = {"y_f16": yf} if d.h16 else {"y_f32": yf}
    if FUSE_DENSE_LN and h16 is None and M >= FUSE_DENSE_LN_MIN_ROWS:
        # one launch: the GEMM's blocks finish the rows of their own stripe (icka_gemm_ln; bitwise the two launches below);
        # shapes the fused kernel does not take (more than one 128-row block per CU: M > 4096 on MI355X) return False
        sync = A._ws.get("gemm_ln_sync")
        if sync is None:
            sync = A._ws["gemm_ln_sync"] = K.gemm_ln_sync(h.device)
        if K.gemm_ln(h, A.w(mod.dense.weight), o, mod.dense.bias, res, mod.LayerNorm.weight, mod.LayerNorm.bias, y, sync,
                     xhat=xhat, rstd=rstd, eps=d.eps, p_drop=d.p_hidden, seed=seed_h, **twin):
            return y, yf, ((xhat, rstd, seed_h) if save else None)
    if h16 is not None:
        K.gemm(K.GEMM_NT, h16, A.w16(mod.dense.weight), o)
    else:
        K.gemm(K.GEMM_NT, h, A.w(mod.dense.weight), o)
    K.ln_fwd(o, mod.dense.bias, res, mod.LayerNorm.weight, mod.LayerNorm.bias, y, xhat=xhat, rstd=rstd,
             eps=d.eps, p_drop=d.p_hidden, seed=seed_h, **twin)
    return y, yf, ((xhat, rstd, seed_h) if save else None)


def _dense_norm_bwd(A: ParamArena, tag: str, mod, h, d: Dims, saved, dy, dy2):
    """Returns (do, dres): gradient of the dense OUTPUT (dropout mask applied; the caller turns it into the gradient of
    ``h`` with the epilogue it wants) and of the residual input.  Queues dW / db (_wgrad) and the LayerNorm parameter
    reduction; the caller flushes."""
    xhat, rstd, seed_h = saved
    M, H = xhat.shape
    dres = _empty(h, M, H)
    do = _empty(h, M, H)
    _ln_bwd_deferred(A, tag, mod.LayerNorm, dy, xhat, rstd, dy2=dy2, dres=dres, dx=do, p_drop=d.p_hidden, seed=seed_h)
    # the dense bias gradient (column sums of do) rides on the weight-gradient GEMM
    _wgrad(A, do, h, A.g(mod.dense.weight), A.grad_beta(mod.dense.weight), bias=mod.dense.bias)
    return do, dres


def _inter_fwd(A: ParamArena, inter, x, x16=None):
    """BertIntermediate (:548-551): gelu(dense(x)); the pre-activation z is the GEMM's second output.  x16 (mixed16): fp16
    operands; the GEMM then writes gelu(z) twice, fp16 (operand of the BertOutput GEMM) and bf16 (operand of its weight
    gradient).  Returns (g bf16, z bf16, g fp16 or None)."""
    M = x.shape[0]
    I = inter.dense.weight.shape[0]
    z = _empty(x, M, I)
    g = _empty(x, M, I)
    if x16 is not None:
        g16 = _empty(x, M, I, dtype=F16)
        K.gemm(K.GEMM_NT, x16, A.w16(inter.dense.weight), g16, bias=inter.dense.bias, epilogue=K.EPI_GELU, out2=z, out3=g)
        return g, z, g16
    K.gemm(K.GEMM_NT, x, A.w(inter.dense.weight), g, bias=inter.dense.bias, epilogue=K.EPI_GELU, out2=z)
    return g, z, None


def _inter_bwd(A: ParamArena, inter, x, dz, dres):
    """dz = gradient of the pre-activation.  Returns dx (+ dres when given)."""
    _wgrad(A, dz, x, A.g(inter.dense.weight), A.grad_beta(inter.dense.weight), bias=inter.dense.bias)
    dx = _empty(x, x.shape[0], x.shape[1])
    if dres is not None:
        K.gemm(K.GEMM_NN, dz, A.w(inter.dense.weight), dx, epilogue=K.EPI_ADD, aux=dres)
    else:
        K.gemm(K.GEMM_NN, dz, A.w(inter.dense.weight), dx)
    return dx


def _attn_block_fwd(A: ParamArena, att, x, xres, kv_src, add_mask, d: Dims, Skv: int, save: bool, kv16=None):
    """BertAttention / BertCrossAttention: projections -> fused attention -> out-proj -> bias+dropout+residual+LN.
    ``att`` is the reference-named module (``.self.{query,key,value}``, ``.output.{dense,LayerNorm}``).
    x is the bf16 MFMA operand; xres its forward twin: f32 (residual only; None -> x) or, in the mixed16 mode, fp16
    (residual AND forward GEMM operand).  Returns (y bf16, y twin, saved)."""
    xres = _fwd_twin(A, x, xres, d)
    ctx, ctx16, s_core = _attn_core_fwd(A, att.self, x, kv_src, add_mask, d, Skv, save, x16=xres if d.h16 else None,
                                        kv16=kv16 if d.h16 else None)
    y, yf, s_out = _dense_norm_fwd(A, att.output, ctx, x if xres is None else xres, d, save, h16=ctx16)
    return y, yf, ((ctx, s_core, s_out) if save else None)


def _attn_block_bwd(A: ParamArena, att, x, kv_src, add_mask, d: Dims, Skv: int, saved, dy, dy2, need_dkv_src: bool):
    """Returns (dx, dkv_src).  dy2 is an optional second gradient of the block output (fused into the LN backward)."""
    ctx, s_core, s_out = saved
    dao, dres = _dense_norm_bwd(A, "ln_attn", att.output, ctx, d, s_out, dy, dy2)
    dctx = _empty(x, x.shape[0], x.shape[1])
    K.gemm(K.GEMM_NN, dao, A.w(att.output.dense.weight), dctx)
    return _attn_core_bwd(A, att.self, x, kv_src, add_mask, d, Skv, s_core, ctx, dctx, dres, need_dkv_src)


def _ffn_block_fwd(A: ParamArena, layer, x, xres, d: Dims, save: bool):
    """BertIntermediate + BertOutput.  Returns (y bf16, y twin, saved)."""
    xres = _fwd_twin(A, x, xres, d)
    g, z, g16 = _inter_fwd(A, layer.intermediate, x, xres if d.h16 else None)
    y, yf, s_out = _dense_norm_fwd(A, layer.output, g, x if xres is None else xres, d, save, h16=g16)
    return y, yf, ((z, g, s_out) if save else None)


def _ffn_block_bwd(A: ParamArena, layer, x, d: Dims, saved, dy, dy2=None):
    """Returns the gradient of the block input: through the dense path (dz @ W1) WITH the residual gradient added."""
    z, g, s_out = saved
    dfo, dres = _dense_norm_bwd(A, "ln_ffn", layer.output, g, d, s_out, dy, dy2)
    dz = _empty(x, z.shape[0], z.shape[1])
    K.gemm(K.GEMM_NN, dfo, A.w(layer.output.dense.weight), dz, epilogue=K.EPI_DGELU, aux=z)
    return _inter_bwd(A, layer.intermediate, x, dz, dres)


# =============================================================================================== Functions
class EmbeddingsFn(torch.autograd.Function):
    """BertEmbeddings.forward (Cross_Modal_Interaction_Module.py:398-412)."""

    @staticmethod
    def forward(ctx, anchor, mod, A: ParamArena, ids, tt, d: Dims):
        B, S = ids.shape
        H = d.H
        save = any(ctx.needs_input_grad)  # grad mode is always off inside Function.forward
        y = torch.empty(B * S, H, dtype=BF16, device=ids.device)
        yf = torch.empty(B * S, H, dtype=F16 if d.h16 else F32, device=ids.device)   # forward twin (see _fwd_twin)
        xhat = torch.empty_like(y) if save else None
        rstd = torch.empty(B * S, dtype=F32, device=ids.device) if save else None
        seed = A.next_seed() if d.p_hidden > 0 else 0
        K.embed_fwd(ids, tt, mod.word_embeddings.weight, mod.position_embeddings.weight,
                    mod.token_type_embeddings.weight, mod.LayerNorm.weight, mod.LayerNorm.bias, y,
                    xhat=xhat, rstd=rstd, eps=d.eps, p_drop=d.p_hidden, seed=seed,
                    **({"y_f16": yf} if d.h16 else {"y_f32": yf}))
        ctx.mod, ctx.A, ctx.d, ctx.seed = mod, A, d, seed
        ctx.save_for_backward(ids, tt, xhat, rstd)
        ctx.mark_non_differentiable(yf)
        ctx.set_materialize_grads(False)   # no 12.6 MB zero-fill for the twin's (unused) gradient
        return y, yf

    @staticmethod
    def backward(ctx, dy, _dyf=None):
        mod, A, d = ctx.mod, ctx.A, ctx.d
        ids, tt, xhat, rstd = ctx.saved_tensors
        dy = _c(dy)
        ws = A.workspace("ln", K._lib.load().icka_ln_bwd_workspace_floats(d.H))
        tables = (mod.word_embeddings.weight, mod.position_embeddings.weight)
        small = (mod.token_type_embeddings.weight, mod.LayerNorm.weight, mod.LayerNorm.bias)
        red = A.reducer
        if red is not None and getattr(red, "sparse_word", None) is A.slot(tables[0]):
            # row-sparse data-parallel exchange of the word-embedding gradient (dp.GradReducer(sparse_embeddings=True)): the
            # kernel leaves the per-token rows, the reducer all-gathers them with the ids and scatters every rank's rows
            beta_w = A.grad_beta(tables[0])
            if A.grad_beta(tables[1]) == 0.0:
                K.zero_(A.g(tables[1]).view(-1))
            acc = A.grad_beta(small) > 0
            dtok = A.workspace("dtok", ids.numel() * d.H).view(ids.numel(), d.H)
            K.embed_bwd_rows(dy, ids, tt, xhat, rstd, mod.LayerNorm.weight, dtok, A.g(tables[1]), A.g(small[0]), A.g(small[1]),
                             A.g(small[2]), ws, vocab=tables[0].shape[0], padding_idx=0, p_drop=d.p_hidden, seed=ctx.seed,
                             accumulate=acc)
            red.set_sparse_rows(dtok, ids.reshape(-1), accumulate=beta_w > 0)
            A.flush_final()
            return None, None, None, None, None, None
        if A.grad_beta(tables) == 0.0:   # atomically accumulated tables: fresh gradient starts from zero
            K.zero_(A.g_cat(tables).view(-1))
        acc = A.grad_beta(small) > 0
        K.embed_bwd(dy, ids, tt, xhat, rstd, mod.LayerNorm.weight, A.g(tables[0]), A.g(tables[1]), A.g(small[0]),
                    A.g(small[1]), A.g(small[2]), ws, padding_idx=0, p_drop=d.p_hidden, seed=ctx.seed, accumulate=acc)
        A.flush_final()
        return None, None, None, None, None, None


class BertLayerFn(torch.autograd.Function):
    """BertLayer.forward (:438-442): self-attention block + feed-forward block."""

    @staticmethod
    def forward(ctx, anchor, x, xf, layer, A: ParamArena, add_mask, d: Dims):
        save = any(ctx.needs_input_grad)  # grad mode is always off inside Function.forward
        x1, x1f, s_att = _attn_block_fwd(A, layer.attention, x, xf, None, add_mask, d, d.S, save)
        x2, x2f, s_ffn = _ffn_block_fwd(A, layer, x1, x1f, d, save)
        ctx.layer, ctx.A, ctx.d, ctx.s_att, ctx.s_ffn = layer, A, d, s_att, s_ffn
        ctx.save_for_backward(x, x1, add_mask)
        ctx.mark_non_differentiable(x2f)
        ctx.set_materialize_grads(False)
        return x2, x2f

    @staticmethod
    def backward(ctx, dy, _dyf=None):
        x, x1, add_mask = ctx.saved_tensors
        layer, A, d = ctx.layer, ctx.A, ctx.d
        dx1 = _ffn_block_bwd(A, layer, x1, d, ctx.s_ffn, _c(dy))
        dx, _ = _attn_block_bwd(A, layer.attention, x, None, add_mask, d, d.S, ctx.s_att, dx1, None, False)
        if not A.keep_saved:
            ctx.s_att = ctx.s_ffn = None
        _flush_wgrad(A)
        A.flush_final()
        return None, dx, None, None, None, None, None


class CrossLayerFn(torch.autograd.Function):
    """BertCrossAttentionLayer.forward (:646-650): Q from s1 (text), K/V from s2 (regions), residual = s1."""

    @staticmethod
    def forward(ctx, anchor, s1, s1f, s2, layer, A: ParamArena, add_mask, d: Dims, s2_16=None):
        """s2_16 (mixed16): the fp16 twin of the key/value source (the projected regions)."""
        save = any(ctx.needs_input_grad)  # grad mode is always off inside Function.forward
        x1, x1f, s_att = _attn_block_fwd(A, layer.attention, s1, s1f, s2, add_mask, d, d.R, save, kv16=s2_16)
        x2, x2f, s_ffn = _ffn_block_fwd(A, layer, x1, x1f, d, save)
        ctx.layer, ctx.A, ctx.d, ctx.s_att, ctx.s_ffn = layer, A, d, s_att, s_ffn
        ctx.need_s2 = s2.requires_grad
        ctx.save_for_backward(s1, s2, x1, add_mask)
        ctx.mark_non_differentiable(x2f)
        ctx.set_materialize_grads(False)
        return x2, x2f

    @staticmethod
    def backward(ctx, dy, _dyf=None):
        s1, s2, x1, add_mask = ctx.saved_tensors
        layer, A, d = ctx.layer, ctx.A, ctx.d
        dx1 = _ffn_block_bwd(A, layer, x1, d, ctx.s_ffn, _c(dy))
        ds1, ds2 = _attn_block_bwd(A, layer.attention, s1, s2, add_mask, d, d.R, ctx.s_att, dx1, None, ctx.need_s2)
        if not A.keep_saved:
            ctx.s_att = ctx.s_ffn = None
        _flush_wgrad(A)
        A.flush_final()
        return None, ds1, None, ds2, None, None, None, None, None


class AttnCoreFn(torch.autograd.Function):
    """BertSelfAttention.forward(hidden_states, attention_mask) (:478-506) / BertCoAttention.forward(s1, s2, s2_mask)
    (:590-624) called on its own: projections + fused attention -> context layer [M,H]."""

    @staticmethod
    def forward(ctx, anchor, x, kv_src, sa, A: ParamArena, add_mask, d: Dims, Skv: int):
        save = any(ctx.needs_input_grad)
        c, _c16, saved = _attn_core_fwd(A, sa, x, kv_src, add_mask, d, Skv, save)
        ctx.sa, ctx.A, ctx.d, ctx.Skv, ctx.saved = sa, A, d, Skv, saved
        ctx.need_kv = kv_src is not None and kv_src.requires_grad
        ctx.save_for_backward(x, kv_src, add_mask, c)
        return c

    @staticmethod
    def backward(ctx, dc):
        x, kv_src, add_mask, c = ctx.saved_tensors
        A = ctx.A
        dx, dsrc = _attn_core_bwd(A, ctx.sa, x, kv_src, add_mask, ctx.d, ctx.Skv, ctx.saved, c, _c(dc), None, ctx.need_kv)
        if not A.keep_saved:
            ctx.saved = None
        _flush_wgrad(A)
        A.flush_final()
        return None, dx, dsrc, None, None, None, None, None


class DenseResidualNormFn(torch.autograd.Function):
    """BertSelfOutput.forward / BertOutput.forward (hidden_states, input_tensor) (:561-565, :532-536):
    LayerNorm(dropout(dense(hidden_states)) + input_tensor).  ``resf`` is the f32 twin of the residual if it has one."""

    @staticmethod
    def forward(ctx, anchor, h, res, resf, mod, A: ParamArena, d: Dims):
        save = any(ctx.needs_input_grad)
        y, yf, saved = _dense_norm_fwd(A, mod, h, res if resf is None else resf, d, save)
        ctx.mod, ctx.A, ctx.d, ctx.saved = mod, A, d, saved
        ctx.save_for_backward(h)
        ctx.mark_non_differentiable(yf)
        ctx.set_materialize_grads(False)
        return y, yf

    @staticmethod
    def backward(ctx, dy, _dyf=None):
        (h,) = ctx.saved_tensors
        mod, A = ctx.mod, ctx.A
        do, dres = _dense_norm_bwd(A, "ln_sub", mod, h, ctx.d, ctx.saved, _c(dy), None)
        dh = torch.empty_like(h)
        K.gemm(K.GEMM_NN, do, A.w(mod.dense.weight), dh)
        if not A.keep_saved:
            ctx.saved = None
        _flush_wgrad(A)
        A.flush_final()
        return None, dh, dres, None, None, None, None


class IntermediateFn(torch.autograd.Function):
    """BertIntermediate.forward (:548-551): gelu(dense(hidden_states))."""

    @staticmethod
    def forward(ctx, anchor, x, inter, A: ParamArena):
        g, z, _g16 = _inter_fwd(A, inter, x)
        ctx.inter, ctx.A = inter, A
        ctx.save_for_backward(x, z)
        return g

    @staticmethod
    def backward(ctx, dg):
        x, z = ctx.saved_tensors
        A = ctx.A
        dz = K.dgelu(_c(dg), z, torch.empty_like(z))
        dx = _inter_bwd(A, ctx.inter, x, dz, None)
        _flush_wgrad(A)
        A.flush_final()
        return None, dx, None, None


class LinearFn(torch.autograd.Function):
    """nn.Linear on a [tokens, in] bf16 matrix (vismap2text :958, generic dense)."""

    @staticmethod
    def forward(ctx, anchor, x, lin, A: ParamArena, out_f32: bool, epilogue: int, x16=None):
        """x16 (mixed16): the fp16 twin of x -> the GEMM reads fp16 operands (x16, the fp16 weight shadow) and returns
        (y bf16, y fp16); the backward is the bf16 one either way."""
        M = x.shape[0]
        N = lin.weight.shape[0]
        if x16 is not None:
            if out_f32 or epilogue != K.EPI_NONE:
                raise ValueError("LinearFn: the fp16-operand form is a plain bf16-output projection")
            if A.shadow16 is None:
                A.enable_fp16_shadow()
                A.sync(force=True)
            y = torch.empty(M, N, dtype=BF16, device=x.device)
            y16 = torch.empty(M, N, dtype=F16, device=x.device)
            K.gemm(K.GEMM_NT, x16, A.w16(lin.weight), y16, bias=lin.bias, out3=y)
            ctx.lin, ctx.A, ctx.epi = lin, A, epilogue
            ctx.need_dx = x.requires_grad
            ctx.save_for_backward(x, None)
            ctx.mark_non_differentiable(y16)
            ctx.set_materialize_grads(False)
            return y, y16
        y = torch.empty(M, N, dtype=F32 if out_f32 else BF16, device=x.device)
        if M <= 64 and not out_f32 and x.shape[1] % 128 == 0 and x.stride(0) % 8 == 0 and \
                epilogue in (K.EPI_NONE, K.EPI_TANH):
            # a handful of rows (the pooler): operands-from-L2 kernel instead of 128x128 tiles on 32 rows
            K.linear_small_m(x, A.w(lin.weight), lin.bias, y, act=1 if epilogue == K.EPI_TANH else 0)
        else:
            K.gemm(K.GEMM_NT, x, A.w(lin.weight), y, bias=lin.bias, epilogue=epilogue)
        ctx.lin, ctx.A, ctx.epi = lin, A, epilogue
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x, y if epilogue == K.EPI_TANH else None)
        return y

    @staticmethod
    def backward(ctx, dy, _dy16=None):
        x, y = ctx.saved_tensors
        lin, A = ctx.lin, ctx.A
        M, N = dy.shape
        if dy.dtype == F32:
            ldd = (N + 7) // 8 * 8
            dyb = torch.empty(M, ldd, dtype=BF16, device=dy.device)
            K.cast_pad_f32_to_bf16(dy if dy.stride(1) == 1 else dy.contiguous(), dyb)
            dyv = dyb[:, :N]
        else:
            dyv = _c(dy)
        if ctx.epi == K.EPI_TANH:   # y = tanh(pre): d(pre) = dy * (1 - y^2)
            dyv = K.tanh_bwd(dyv if dyv.is_contiguous() else dyv.contiguous(), y, torch.empty_like(y))
        fused = lin.bias is not None and N % 128 == 0 and x.shape[1] % 128 == 0 and M % 64 == 0
        gw = A.g(lin.weight)
        beta_w = A.grad_beta(lin.weight)
        # + the data-parallel wire copy -- for outputs on the aligned fast path only: a skinny gradient (the classifier's
        # 13 x 768 over K = 4096 tokens) keeps its split-K form, which a wire copy would switch off (gemm.hip: the copy needs
        # the final value in one epilogue); the bucket's chunk cast covers such a slot
        wire = A.wire_of(gw, beta_w) if (N % 128 == 0 and x.shape[1] % 128 == 0) else {}
        K.gemm(K.GEMM_TN, dyv, x, gw, beta=beta_w,
               colsum_out=A.g(lin.bias) if fused else None,
               colsum_accumulate=fused and A.grad_beta(lin.bias) > 0,   # bias gradient inside the wgrad GEMM
               **wire)
        if lin.bias is not None and not fused:
            csw = A.workspace("colsum", K._lib.load().icka_colsum_workspace_floats(N))
            K.colsum(dyv, A.g(lin.bias), csw, accumulate=A.grad_beta(lin.bias) > 0)
        dx = None
        if ctx.need_dx:
            dx = torch.empty_like(x)
            K.gemm(K.GEMM_NN, dyv, A.w(lin.weight), dx)
        A.flush_final()
        return None, dx, None, None, None, None, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on the encoder output (:953); writes optional second copy (concat buffer)."""

    @staticmethod
    def forward(ctx, x, A: ParamArena, p: float):
        seed = A.next_seed()
        y = torch.empty_like(x)
        K.dropout(x, y, p_drop=p, seed=seed)
        ctx.p, ctx.seed = p, seed
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        K.dropout(dy, dx, p_drop=ctx.p, seed=ctx.seed)
        return dx, None, None


class GatedHeadFn(torch.autograd.Function):
    """my_bert/cl_modeling.py:1363-1371:  Gate = sigmoid(Gate_text(seq) + Gate_image(cross));
    logits = classifier(cat(seq, Gate * cross)).  Neither concat is materialised: the gate GEMM reads its K
    reduction from two buffers / two weight matrices, and so does the classifier GEMM."""

    @staticmethod
    def forward(ctx, anchor, seq, cross, head, A: ParamArena, seq16=None, cross16=None):
        """seq16 / cross16 (mixed16): the fp16 twins of seq / cross.  The gate GEMM then reads fp16 operands and writes the
        gated product in fp16 (+ the bf16 copy the backward reads), and the classifier reads fp16 inputs and weights."""
        M, H = seq.shape
        C = head.classifier.weight.shape[0]
        gate = torch.empty(M, H, dtype=BF16, device=seq.device)
        gated = torch.empty_like(gate)
        h16 = seq16 is not None and cross16 is not None
        ctx.skinny = C <= 16 and H % 8 == 0 and 2 * H // 8 <= 256 and C * 2 * H * 2 <= 44 * 1024
        logits = torch.empty(M, C, dtype=F32, device=seq.device)
        if h16:
            if A.shadow16 is None:
                A.enable_fp16_shadow()
                A.sync(force=True)
            gated16 = torch.empty(M, H, dtype=F16, device=seq.device)
            K.gemm(K.GEMM_NT, seq16, A.w16(head.Gate_text.weight), gated16, A2=cross16, B2=A.w16(head.Gate_image.weight),
                   bias=head.Gate_text.bias, bias2=head.Gate_image.bias, epilogue=K.EPI_GATE, aux=cross16, out2=gate,
                   out3=gated)
            Wc = A.w16(head.classifier.weight)
            if ctx.skinny:
                K.cls_head_fwd(seq16, gated16, Wc, head.classifier.bias, logits)
            else:
                K.gemm(K.GEMM_NT, seq16, Wc[:, :H], logits, A2=gated16, B2=Wc[:, H:], bias=head.classifier.bias)
        else:
            K.gemm(K.GEMM_NT, seq, A.w(head.Gate_text.weight), gated, A2=cross, B2=A.w(head.Gate_image.weight),
                   bias=head.Gate_text.bias, bias2=head.Gate_image.bias, epilogue=K.EPI_GATE, aux=cross, out2=gate)
            Wc = A.w(head.classifier.weight)
            if ctx.skinny:   # HBM-bound kernels for the 13-wide output (see icka_cls_head_fwd)
                K.cls_head_fwd(seq, gated, Wc, head.classifier.bias, logits)
            else:
                K.gemm(K.GEMM_NT, seq, Wc[:, :H], logits, A2=gated, B2=Wc[:, H:], bias=head.classifier.bias)
        ctx.head, ctx.A = head, A
        ctx.save_for_backward(seq, cross, gate, gated)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        seq, cross, gate, gated = ctx.saved_tensors
        head, A = ctx.head, ctx.A
        M, H = seq.shape
        C = dlogits.shape[1]
        if dlogits.dtype == BF16 and dlogits.stride(1) == 1 and dlogits.stride(0) % 8 == 0:
            dl = dlogits
        else:
            buf = torch.empty(M, (C + 7) // 8 * 8, dtype=BF16, device=seq.device)
            K.cast_pad_f32_to_bf16(dlogits.float() if dlogits.dtype != F32 else
                                   (dlogits if dlogits.stride(1) == 1 else dlogits.contiguous()), buf)
            dl = buf[:, :C]
        cls = head.classifier
        if ctx.skinny:
            # classifier dgrad + gate backward + dW/db slabs in one launch; the slabs are summed by extra blocks of
            # the gate weight-gradient launch below
            lib = K._lib.load()
            nslab, sf = lib.icka_cls_head_bwd_slabs(M), lib.icka_cls_head_slab_floats(H, C)
            ws = A.workspace("cls_head", nslab * sf)
            dseq_c = torch.empty(M, H, dtype=BF16, device=seq.device)
            du = torch.empty_like(dseq_c)
            dcross_d = torch.empty_like(dseq_c)
            K.cls_head_bwd(dl, seq, gated, gate, cross, A.w(cls.weight), dseq_c, du, dcross_d, ws)
            A.pending_reductions.append(K.slab_reduction(ws, nslab, C * 2 * H, (A.g(cls.weight).view(-1),),
                                                         A.grad_beta(cls.weight) > 0, slab_stride=sf))
            A.pending_reductions.append(K.slab_reduction(ws, nslab, C, (A.g(cls.bias),), A.grad_beta(cls.bias) > 0,
                                                         slab_stride=sf, offset=C * 2 * H))
            return GatedHeadFn._gate_tail(A, head, seq, cross, du, dseq_c, dcross_d)
        gW = A.g(cls.weight)
        beta = A.grad_beta(cls.weight)
        K.gemm(K.GEMM_TN, dl, seq, gW[:, :H], beta=beta)
        K.gemm(K.GEMM_TN, dl, gated, gW[:, H:], beta=beta)
        csw = A.workspace("colsum", K._lib.load().icka_colsum_workspace_floats(max(H, C)))
        K.colsum(dl, A.g(cls.bias), csw, accumulate=A.grad_beta(cls.bias) > 0)
        Wc = A.w(cls.weight)
        dseq_c = torch.empty(M, H, dtype=BF16, device=seq.device)
        dgated = torch.empty_like(dseq_c)
        K.gemm(K.GEMM_NN, dl, Wc[:, :H], dseq_c)
        K.gemm(K.GEMM_NN, dl, Wc[:, H:], dgated)
        du = torch.empty_like(dseq_c)
        dcross_d = torch.empty_like(dseq_c)
        K.gate_bwd(dgated, gate, cross, du, dcross_d)
        return GatedHeadFn._gate_tail(A, head, seq, cross, du, dseq_c, dcross_d)

    @staticmethod
    def _gate_tail(A, head, seq, cross, du, dseq_c, dcross_d):
        # both gate weight gradients (and their bias gradients = column sums of du) in one grouped launch
        _wgrad(A, du, seq, A.g(head.Gate_text.weight), A.grad_beta(head.Gate_text.weight), bias=head.Gate_text.bias)
        _wgrad(A, du, cross, A.g(head.Gate_image.weight), A.grad_beta(head.Gate_image.weight),
               bias=head.Gate_image.bias)
        _flush_wgrad(A)
        dseq = torch.empty_like(dseq_c)
        dcross = torch.empty_like(dseq_c)
        K.gemm(K.GEMM_NN, du, A.w(head.Gate_text.weight), dseq, epilogue=K.EPI_ADD, aux=dseq_c)
        K.gemm(K.GEMM_NN, du, A.w(head.Gate_image.weight), dcross, epilogue=K.EPI_ADD, aux=dcross_d)
        A.flush_final()
        return None, dseq, dcross, None, None, None, None


class SampleGateFn(torch.autograd.Function):
    """Per-sample gates (kernels: fusion.hip).  mode 0: out = g*a + (1-g)*c with g = sigmoid(gate[b])
    (Cross_Modal_Interaction_Module.py:1035-1036); mode 1: out = softmax(gate[b])[1] * a (gate_cl_modeling.py:1369-1373)."""

    @staticmethod
    def forward(ctx, a, c, gate, mode: int, B: int, S: int, a16=None):
        """a16 (mixed16, no blend operand): the fp16 twin of a -> returns (out bf16, out fp16), both from g * a16."""
        out = torch.empty_like(a)
        ctx.mode, ctx.B, ctx.S = mode, B, S
        ctx.save_for_backward(a, c, gate)
        if a16 is not None:
            if c is not None:
                raise ValueError("SampleGateFn: the fp16 form has no blend operand")
            out16 = torch.empty(a.shape[0], a.shape[1], dtype=F16, device=a.device)
            K.sample_gate_fwd_h(a16, gate, mode, out, out16, B, S)
            ctx.mark_non_differentiable(out16)
            ctx.set_materialize_grads(False)
            return out, out16
        K.sample_gate_fwd(a, c, gate, mode, out, B, S)
        return out

    @staticmethod
    def backward(ctx, dout, _d16=None):
        a, c, gate = ctx.saved_tensors
        dout = _c(dout)
        da = torch.empty_like(a)
        dc = torch.empty_like(a) if c is not None else None
        dgate = torch.zeros_like(gate)
        K.sample_gate_bwd(dout, a, c, gate, ctx.mode, da, dc, dgate, ctx.B, ctx.S)
        return da, dc, dgate, None, None, None, None


class CrsFn(torch.autograd.Function):
    """crs_classifier(cat(seq, cross).view(B, -1)) of gate_cl (gate_cl_modeling.py:1364-1366) -> f32 [B,2]."""

    @staticmethod
    def forward(ctx, anchor, seq, cross, lin, A: ParamArena, B: int, S: int):
        crs = torch.zeros(B, 2, dtype=F32, device=seq.device)
        K.crs_fwd(seq, cross, A.w(lin.weight), lin.bias, crs, B, S)
        ctx.lin, ctx.A, ctx.B, ctx.S = lin, A, B, S
        ctx.save_for_backward(seq, cross)
        return crs

    @staticmethod
    def backward(ctx, dcrs):
        seq, cross = ctx.saved_tensors
        lin, A = ctx.lin, ctx.A
        dseq = torch.empty(seq.shape[0], seq.shape[1], dtype=BF16, device=seq.device)
        dcross = torch.empty_like(dseq)
        acc = A.grad_beta((lin.weight, lin.bias)) > 0
        K.crs_bwd(dcrs.contiguous(), seq, cross, A.w(lin.weight), dseq, dcross, A.g(lin.weight), A.g(lin.bias),
                  ctx.B, ctx.S, acc)
        A.flush_final()
        return None, dseq, dcross, None, None, None, None


class AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(x + r) on [rows, H] (cls_layer_both.forward, Cross_Modal_Interaction_Module.py:879-884)."""

    @staticmethod
    def forward(ctx, anchor, x, r, norm, A: ParamArena, eps: float):
        M, H = x.shape
        y = torch.empty(M, H, dtype=BF16, device=x.device)
        xhat = torch.empty_like(y)
        rstd = torch.empty(M, dtype=F32, device=x.device)
        K.ln_fwd(x, None, r, norm.weight, norm.bias, y, xhat=xhat, rstd=rstd, eps=eps)
        ctx.norm, ctx.A = norm, A
        ctx.save_for_backward(xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd = ctx.saved_tensors
        norm, A = ctx.norm, ctx.A
        dy = _c(dy)
        d = torch.empty_like(dy)
        ws = A.workspace("ln", K._lib.load().icka_ln_bwd_workspace_floats(dy.shape[1]))
        acc = A.grad_beta((norm.weight, norm.bias)) > 0
        K.ln_bwd(dy, xhat, rstd, norm.weight, dres=d, dgamma=A.g(norm.weight), dbeta=A.g(norm.bias), partials=ws,
                 accumulate=acc)
        A.flush_final()
        return None, d, d, None, None, None


class FanOutFn(torch.autograd.Function):
    """y1 = y2 = ... = x for a tensor with several consumers; the backward sums the branch gradients with the bf16
    add kernel instead of leaving the fan-in to autograd's own accumulation."""

    @staticmethod
    def forward(ctx, x, n: int):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *grads):
        acc = None
        for g in grads:
            if g is None:
                continue
            g = _c(g)
            acc = g if acc is None else K.add_bf16(acc, g, torch.empty_like(g))
        return acc, None


class PromptMappingFn(torch.autograd.Function):
    """The reference's prompt mapping networks (Cross_Modal_Interaction_Module.py:914-928, calls :995, :998):
    nn.Sequential(Dropout(p), Linear(in, N1), Tanh(), Dropout(p), Linear(N1, N2)) on [rows, in] bf16.
    N1 = 756 * prompt_len = 3780 is not a multiple of 8, so the hidden activation lives in a zero-padded
    [rows, N1p] buffer (N1p = N1 rounded up to 8): dropout / tanh' run on the padded rows, the GEMMs on views."""

    @staticmethod
    def forward(ctx, anchor, x, lin1, lin2, A: ParamArena, p: float):
        M, Kin = x.shape
        N1, N2 = lin1.weight.shape[0], lin2.weight.shape[0]
        N1p = (N1 + 7) // 8 * 8
        s1 = A.next_seed() if p > 0 else 0
        s2 = A.next_seed() if p > 0 else 0
        xd = x
        if p > 0:
            xd = torch.empty_like(x)
            K.dropout(x, xd, p_drop=p, seed=s1)
        h = torch.zeros(M, N1p, dtype=BF16, device=x.device)
        K.gemm(K.GEMM_NT, xd, A.w(lin1.weight), h[:, :N1], bias=lin1.bias, epilogue=K.EPI_TANH)
        hd = h
        if p > 0:
            hd = torch.empty_like(h)
            K.dropout(h, hd, p_drop=p, seed=s2)
        y = torch.empty(M, N2, dtype=BF16, device=x.device)
        # W2 [N2, N1] has 7560-byte rows in the arena (not 16-byte aligned): a zero-padded bf16 copy [N2, N1p] keeps the
        # operand loads of this GEMM and of d(hidden) = dy . W2 vectorised (one cast launch from the f32 master)
        w2p = K.cast_pad_f32_to_bf16(lin2.weight.detach(), torch.empty(N2, N1p, dtype=BF16, device=x.device)) \
            if N1p != N1 else A.w(lin2.weight)
        K.gemm(K.GEMM_NT, hd, w2p, y, bias=lin2.bias)
        ctx.lin1, ctx.lin2, ctx.A, ctx.p, ctx.s1, ctx.s2 = lin1, lin2, A, p, s1, s2
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(xd, h, hd, w2p)
        return y

    @staticmethod
    def backward(ctx, dy):
        xd, h, hd, w2p = ctx.saved_tensors
        lin1, lin2, A, p = ctx.lin1, ctx.lin2, ctx.A, ctx.p
        dy = _c(dy)
        M, N2 = dy.shape
        N1, N1p = lin1.weight.shape[0], h.shape[1]
        lib = K._lib.load()
        # second Linear: dW2 = dy^T . hd, db2 = colsum(dy), d(hd) = dy . W2
        K.gemm(K.GEMM_TN, dy, hd[:, :N1], A.g(lin2.weight), beta=A.grad_beta(lin2.weight))
        K.colsum(dy, A.g(lin2.bias), A.workspace("colsum", lib.icka_colsum_workspace_floats(N2)),
                 accumulate=A.grad_beta(lin2.bias) > 0)
        dh = torch.empty(M, N1p, dtype=BF16, device=dy.device)
        K.gemm(K.GEMM_NN, dy, w2p, dh)            # pad columns: dy . 0 = 0
        if p > 0:
            K.dropout(dh, dh, p_drop=p, seed=ctx.s2)
        dpre = K.tanh_bwd(dh, h, torch.empty_like(h))
        # first Linear
        K.gemm(K.GEMM_TN, dpre[:, :N1], xd, A.g(lin1.weight), beta=A.grad_beta(lin1.weight))
        K.colsum(dpre[:, :N1], A.g(lin1.bias), A.workspace("colsum", lib.icka_colsum_workspace_floats(N1)),
                 accumulate=A.grad_beta(lin1.bias) > 0)
        dx = None
        if ctx.need_dx:
            dx = torch.empty_like(xd)
            K.gemm(K.GEMM_NN, dpre[:, :N1], A.w(lin1.weight), dx)
            if p > 0:
                K.dropout(dx, dx, p_drop=p, seed=ctx.s1)
        A.flush_final()
        return None, dx, None, None, None, None


class PromptEmbeddingsFn(torch.autograd.Function):
    """Embeddings of the prompt-accepting encoder stage (icka_hip.h: icka_embed_prompt_fwd): word rows gathered by
    ``src`` with the prompt vectors spliced in, + position (offset) + type row 0 -> LayerNorm -> dropout."""

    @staticmethod
    def forward(ctx, anchor, prompt, mod, A: ParamArena, ids, src, d: Dims, pos_offset: int):
        B = ids.shape[0]
        S, H = src.shape[0], d.H
        save = any(ctx.needs_input_grad)
        y = torch.empty(B * S, H, dtype=BF16, device=ids.device)
        yf = torch.empty(B * S, H, dtype=F32, device=ids.device)
        xhat = torch.empty_like(y) if save else None
        rstd = torch.empty(B * S, dtype=F32, device=ids.device) if save else None
        seed = A.next_seed() if d.p_hidden > 0 else 0
        prompt = prompt if prompt.is_contiguous() else prompt.contiguous()
        K.embed_prompt_fwd(ids, src, prompt, mod.word_embeddings.weight, mod.position_embeddings.weight,
                           mod.token_type_embeddings.weight, mod.LayerNorm.weight, mod.LayerNorm.bias, y, y_f32=yf,
                           xhat=xhat, rstd=rstd, pos_offset=pos_offset, eps=d.eps, p_drop=d.p_hidden, seed=seed)
        ctx.mod, ctx.A, ctx.d, ctx.seed, ctx.pos_offset, ctx.pshape = mod, A, d, seed, pos_offset, tuple(prompt.shape)
        ctx.save_for_backward(ids, src, xhat, rstd)
        ctx.mark_non_differentiable(yf)
        ctx.set_materialize_grads(False)
        return y, yf

    @staticmethod
    def backward(ctx, dy, _dyf=None):
        mod, A, d = ctx.mod, ctx.A, ctx.d
        ids, src, xhat, rstd = ctx.saved_tensors
        dy = _c(dy)
        S = src.shape[0]
        ws = A.workspace("embp", S * K._lib.load().icka_ln_slab_slots() * d.H)
        tables = (mod.word_embeddings.weight, mod.position_embeddings.weight)
        if A.grad_beta(tables) == 0.0:
            K.zero_(A.g_cat(tables).view(-1))
        small = (mod.token_type_embeddings.weight, mod.LayerNorm.weight, mod.LayerNorm.bias)
        acc = A.grad_beta(small) > 0
        dprompt = torch.empty(ctx.pshape, dtype=BF16, device=dy.device)
        pad = mod.word_embeddings.padding_idx
        K.embed_prompt_bwd(dy, ids, src, xhat, rstd, mod.LayerNorm.weight, A.g(tables[0]), A.g(tables[1]),
                           A.g(small[0]), A.g(small[1]), A.g(small[2]), dprompt, ws, pos_offset=ctx.pos_offset,
                           padding_idx=-1 if pad is None else pad, p_drop=d.p_hidden, seed=ctx.seed, accumulate=acc)
        A.flush_final()
        return None, dprompt, None, None, None, None, None, None


class TokenCEFn(torch.autograd.Function):
    """Benchmark loss: token-level cross-entropy, mean over valid tokens.  Two launches (per-block
    partials, finalize; no accumulator fill, no atomics) yield the loss sum, the token count, their ratio and the unscaled
    logit gradient; dloss / #valid is applied on device in
    the backward (no host sync).  The logits are f32, so their gradient is handed to autograd in f32 (a bf16 gradient
    would be cast back by the engine with an ATen kernel); the classifier backward packs it to bf16 itself."""

    @staticmethod
    def forward(ctx, logits, labels, mask):
        M, C = logits.shape
        stats = torch.empty(3, dtype=F32, device=logits.device)
        dl = torch.empty(M, C, dtype=F32, device=logits.device)
        K.token_ce_fused(logits, labels.reshape(-1), mask.reshape(-1), stats, dl)   # sum, count, mean
        ctx.save_for_backward(dl, stats)
        return stats[2:3].view(())

    @staticmethod
    def backward(ctx, dloss):
        dl, stats = ctx.saved_tensors
        g = dloss.reshape(1)
        if g.dtype != F32:
            raise TypeError("loss gradient must be f32")
        out = torch.empty_like(dl)
        check_rc = K._lib.load().icka_x_scale_by_ratio(dl.data_ptr(), out.data_ptr(), g.data_ptr(), stats[1:2].data_ptr(),
                                                      dl.numel(), K._stream())
        K.check(check_rc, "icka_x_scale_by_ratio")
        return out, None, None
