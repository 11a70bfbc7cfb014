"""fp32 "exact" arithmetic mode of the hot path (csrc/exact.hip, include/icka_hip.h: icka_x_*).

BASELINE.json's north_star: logits "within 1e-3 fp32 / 2e-2 bf16" of the reference, which is fp32 end to end (SURVEY.md
section 0; Cross_Modal_Interaction_Module.py:950).  The product kernels are bf16-MFMA and meet the 2e-2 bar; this module
is the same path with f32 activations, the f32 master parameters and f32 arithmetic, selected per model with
``icka_amd.set_precision(model, "fp32")``:

  * every nn.Linear / matmul and its two gradients is ``icka_x_gemm`` (v_mfma_f32_16x16x4_f32, batched over
    (batch, head) for QK^T / PV, any shape -- so this mode has no head-size restriction);
  * attention materialises its probabilities exactly like the reference (:488-502) and saves them for backward;
  * LayerNorm / softmax / erf-GELU / dropout / embeddings are f32 kernels; parameter gradients go straight into the
    ParamArena's f32 gradient buffer like in the bf16 path.

One autograd.Function per reference block, mirroring ``ops.py``.  It is a validation mode (about 1/16 of the bf16 MFMA
rate, unfused): nothing in the bf16 path routes through it and it is never a fallback.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Sequence

import torch

from . import _lib
from ._lib import GEMM_NN, GEMM_NT, GEMM_TN, GEMM_TT, XGemmDesc, check
from .arena import ParamArena

F32 = torch.float32
ACT_GELU, ACT_TANH, ACT_GATE = 0, 1, 2


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f32(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise TypeError("%s must be a ROCm device tensor (icka_amd has no CPU path), got %s" % (name, t.device))
    if t.dtype != F32:
        raise TypeError("%s must be f32 in the fp32-exact mode, got %s" % (name, t.dtype))


def _mat(t: torch.Tensor, name: str) -> None:
    _f32(t, name)
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be a 2-D row-major view (stride(1)==1), got shape %s strides %s"
                         % (name, tuple(t.shape), t.stride()))


def _cf(t: torch.Tensor) -> torch.Tensor:
    """autograd may hand over strided / broadcast gradients: plain contiguous f32."""
    if t.dtype != F32:
        raise TypeError("expected an f32 gradient in the fp32-exact mode, got %s" % t.dtype)
    return t if t.is_contiguous() else t.contiguous()


def _new(ref: torch.Tensor, *shape) -> torch.Tensor:
    return torch.empty(*shape, dtype=F32, device=ref.device)


# =============================================================================================== launchers
def gemm(op: int, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
         alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
    """out[M,N] = alpha * op(A,B) (+ bias) (+ beta*out) on 2-D f32 row-major views."""
    _mat(A, "A"); _mat(B, "B"); _mat(out, "out")
    if op == GEMM_NT:
        (M, K), (N, Kb) = A.shape, B.shape
    elif op == GEMM_NN:
        (M, K), (Kb, N) = A.shape, B.shape
    elif op == GEMM_TN:
        (K, M), (Kb, N) = A.shape, B.shape
    elif op == GEMM_TT:
        (K, M), (N, Kb) = A.shape, B.shape
    else:
        raise ValueError("bad op")
    if K != Kb or tuple(out.shape) != (M, N):
        raise ValueError("shape mismatch: op %d, A %s, B %s, out %s" % (op, tuple(A.shape), tuple(B.shape), tuple(out.shape)))
    return gemm_raw(op, M, N, K, A, A.stride(0), (0, 0), B, B.stride(0), (0, 0), out, out.stride(0), (0, 0), 1, 1,
                    bias=bias, alpha=alpha, beta=beta)


def gemm_raw(op, M, N, K, A, lda, a_bs, B, ldb, b_bs, out, ldc, c_bs, nb0, nb1, *, bias=None, alpha=1.0, beta=0.0):
    """Batched form: X[b0,b1] = X.data_ptr() + (b0*x_bs[0] + b1*x_bs[1]) elements.  The caller vouches for the extents."""
    for n, t in (("A", A), ("B", B), ("out", out)):
        _f32(t, n)
    d = XGemmDesc()
    d.op, d.M, d.N, d.K = op, M, N, K
    d.A, d.lda, d.a_bs0, d.a_bs1 = A.data_ptr(), lda, a_bs[0], a_bs[1]
    d.B, d.ldb, d.b_bs0, d.b_bs1 = B.data_ptr(), ldb, b_bs[0], b_bs[1]
    d.C, d.ldc, d.c_bs0, d.c_bs1 = out.data_ptr(), ldc, c_bs[0], c_bs[1]
    d.nb0, d.nb1 = nb0, nb1
    if bias is not None:
        _f32(bias, "bias")
        if bias.numel() != N or not bias.is_contiguous():
            raise ValueError("bias must be contiguous f32 [N]")
    d.bias = None if bias is None else bias.data_ptr()
    d.alpha, d.beta = alpha, beta
    check(_lib.load().icka_x_gemm(C.byref(d), _stream()), "icka_x_gemm")
    return out


def ln_fwd(x, residual, gamma, beta, *, eps, p_drop=0.0, seed=0, save=True):
    _mat(x, "x")
    M, H = x.shape
    if residual is not None:
        _mat(residual, "residual")
    y = _new(x, M, H)
    xhat = _new(x, M, H) if save else None
    rstd = _new(x, M) if save else None
    check(_lib.load().icka_x_ln_fwd(x.data_ptr(), x.stride(0), None if residual is None else residual.data_ptr(),
                                    0 if residual is None else residual.stride(0), gamma.data_ptr(), beta.data_ptr(),
                                    y.data_ptr(), None if xhat is None else xhat.data_ptr(),
                                    None if rstd is None else rstd.data_ptr(), M, H, eps, p_drop, seed, _stream()),
          "icka_x_ln_fwd")
    return y, xhat, rstd


def ln_bwd(dy, xhat, rstd, gamma, *, p_drop=0.0, seed=0):
    """Returns (dpre, ddense): gradient of the LayerNorm input, and the same times the dropout mask (the dense branch);
    ddense is dpre itself when no dropout was applied."""
    _mat(dy, "dy")
    M, H = dy.shape
    dpre = _new(dy, M, H)
    dd = _new(dy, M, H) if p_drop > 0 else None
    check(_lib.load().icka_x_ln_bwd(dy.data_ptr(), dy.stride(0), xhat.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                    dpre.data_ptr(), None if dd is None else dd.data_ptr(), M, H, p_drop, seed, _stream()),
          "icka_x_ln_bwd")
    return dpre, (dpre if dd is None else dd)


def colsum(a, out, *, b=None, accumulate=False):
    """out[c] (+)= sum_r a[r,c] * (b[r,c] if b is given else 1)."""
    _mat(a, "a")
    M, N = a.shape
    if b is not None:
        _mat(b, "b")
    _f32(out, "out")
    if out.numel() != N or not out.is_contiguous():
        raise ValueError("colsum output must be contiguous f32 [%d]" % N)
    check(_lib.load().icka_x_colsum(a.data_ptr(), a.stride(0), None if b is None else b.data_ptr(),
                                    0 if b is None else b.stride(0), out.data_ptr(), M, N, int(accumulate), _stream()),
          "icka_x_colsum")


def act_fwd(x, mode, *, aux=None, y2=None, out=None):
    _f32(x, "x")
    if not x.is_contiguous():
        raise ValueError("act_fwd: contiguous input")
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().icka_x_act_fwd(x.data_ptr(), None if aux is None else aux.data_ptr(), y.data_ptr(),
                                     None if y2 is None else y2.data_ptr(), x.numel(), mode, _stream()), "icka_x_act_fwd")
    return y


def act_bwd(dy, saved, mode, *, aux=None, dx2=None):
    dy = _cf(dy)
    dx = torch.empty_like(dy)
    check(_lib.load().icka_x_act_bwd(dy.data_ptr(), saved.data_ptr(), None if aux is None else aux.data_ptr(), dx.data_ptr(),
                                     None if dx2 is None else dx2.data_ptr(), dy.numel(), mode, _stream()), "icka_x_act_bwd")
    return dx


def dropout(x, p_drop, seed):
    x = _cf(x)
    y = torch.empty_like(x)
    check(_lib.load().icka_x_dropout(x.data_ptr(), y.data_ptr(), x.numel(), p_drop, seed, _stream()), "icka_x_dropout")
    return y


def concat2(a, b):
    _mat(a, "a"); _mat(b, "b")
    M = a.shape[0]
    out = _new(a, M, a.shape[1] + b.shape[1])
    check(_lib.load().icka_x_concat2(a.data_ptr(), a.stride(0), a.shape[1], b.data_ptr(), b.stride(0), b.shape[1],
                                     out.data_ptr(), M, _stream()), "icka_x_concat2")
    return out


def softmax_fwd(P, Pd, add_mask, B, heads, Sq, Skv, scale, p_drop, seed):
    check(_lib.load().icka_x_softmax_fwd(P.data_ptr(), None if Pd is None else Pd.data_ptr(), add_mask.data_ptr(), B, heads,
                                         Sq, Skv, scale, p_drop, seed, _stream()), "icka_x_softmax_fwd")


def softmax_bwd(P, dS, B, heads, Sq, Skv, scale, p_drop, seed):
    check(_lib.load().icka_x_softmax_bwd(P.data_ptr(), dS.data_ptr(), B, heads, Sq, Skv, scale, p_drop, seed, _stream()),
          "icka_x_softmax_bwd")


# =============================================================================================== building blocks
def _fw(A: ParamArena, ps) -> torch.Tensor:
    """f32 master view of one 2-D parameter, or of several adjacent ones as one [sum rows, in] matrix."""
    if isinstance(ps, torch.nn.Parameter):
        ps = (ps,)
    first, rows = A._adjacent(ps)
    return A.flat[first.off:first.off + rows * first.shape[-1]].view(rows, first.shape[-1])


def _fb(A: ParamArena, ps) -> torch.Tensor:
    if isinstance(ps, torch.nn.Parameter):
        ps = (ps,)
    return A.f_cat(ps)


def _linear_bwd_params(A: ParamArena, dy, x, ws, bs) -> None:
    """dW (+)= dy^T . x ; db (+)= colsum(dy) into the gradient arena (ws / bs: parameter or tuple of adjacent ones)."""
    wt = ws if isinstance(ws, tuple) else (ws,)
    gemm(GEMM_TN, dy, x, A.g_cat(wt), beta=A.grad_beta(wt))
    if bs is not None:
        bt = bs if isinstance(bs, tuple) else (bs,)
        colsum(dy, A.g_cat(bt), accumulate=A.grad_beta(bt) > 0)


def _attn_core_fwd(A: ParamArena, sa, x, kv_src, add_mask, d, Skv: int, save: bool):
    """BertSelfAttention / BertCoAttention (:478-506, :590-624).  Returns (ctx [M,H], saved)."""
    M, H = x.shape
    B, h, S = d.B, d.heads, d.S
    dh = H // h
    if kv_src is None:
        wq = (sa.query.weight, sa.key.weight, sa.value.weight)
        qkv = gemm(GEMM_NT, x, _fw(A, wq), _new(x, M, 3 * H), bias=_fb(A, (sa.query.bias, sa.key.bias, sa.value.bias)))
        q, k, v, kvb = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], None
    else:
        qkv = gemm(GEMM_NT, x, _fw(A, sa.query.weight), _new(x, M, H), bias=sa.query.bias)
        kvb = gemm(GEMM_NT, kv_src, _fw(A, (sa.key.weight, sa.value.weight)), _new(x, kv_src.shape[0], 2 * H),
                   bias=_fb(A, (sa.key.bias, sa.value.bias)))
        q, k, v = qkv, kvb[:, :H], kvb[:, H:]
    P = _new(x, B, h, S, Skv)
    # scores = Q . K^T per (batch, head)                                                          (:488)
    gemm_raw(GEMM_NT, S, Skv, dh, q, q.stride(0), (S * q.stride(0), dh), k, k.stride(0), (Skv * k.stride(0), dh),
             P, Skv, (h * S * Skv, S * Skv), B, h)
    seed_a = A.next_seed() if d.p_attn > 0 else 0
    Pd = torch.empty_like(P) if d.p_attn > 0 else None
    softmax_fwd(P, Pd, add_mask, B, h, S, Skv, 1.0 / math.sqrt(dh), d.p_attn, seed_a)             # (:489-500)
    Pu = P if Pd is None else Pd
    ctx = _new(x, M, H)
    gemm_raw(GEMM_NN, S, dh, Skv, Pu, Skv, (h * S * Skv, S * Skv), v, v.stride(0), (Skv * v.stride(0), dh),
             ctx, H, (S * H, dh), B, h)                                                           # (:502-505)
    return ctx, ((qkv, kvb, P, Pd, seed_a) if save else None)


def _attn_core_bwd(A: ParamArena, sa, x, kv_src, d, Skv: int, saved, dctx, dres, need_dkv_src: bool):
    """Returns (dx, dkv_src); ``dres`` (optional, overwritten) receives the result: dx = dres + dqkv . W."""
    qkv, kvb, P, Pd, seed_a = saved
    M, H = x.shape
    B, h, S = d.B, d.heads, d.S
    dh = H // h
    scale = 1.0 / math.sqrt(dh)
    if kv_src is None:
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        dqkv = _new(x, M, 3 * H)
        dq, dk, dv = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]
    else:
        q, k, v = qkv, kvb[:, :H], kvb[:, H:]
        dq = _new(x, M, H)
        dkv = _new(x, kv_src.shape[0], 2 * H)
        dk, dv = dkv[:, :H], dkv[:, H:]
    pb = (h * S * Skv, S * Skv)
    Pu = P if Pd is None else Pd
    # dPd = dctx . V^T ; dV = Pd^T . dctx
    dS = torch.empty_like(P)
    gemm_raw(GEMM_NT, S, Skv, dh, dctx, H, (S * H, dh), v, v.stride(0), (Skv * v.stride(0), dh), dS, Skv, pb, B, h)
    gemm_raw(GEMM_TN, Skv, dh, S, Pu, Skv, pb, dctx, H, (S * H, dh), dv, dv.stride(0), (Skv * dv.stride(0), dh), B, h)
    softmax_bwd(P, dS, B, h, S, Skv, scale, d.p_attn, seed_a)
    # dQ = dS . K ; dK = dS^T . Q
    gemm_raw(GEMM_NN, S, dh, Skv, dS, Skv, pb, k, k.stride(0), (Skv * k.stride(0), dh), dq, dq.stride(0),
             (S * dq.stride(0), dh), B, h)
    gemm_raw(GEMM_TN, Skv, dh, S, dS, Skv, pb, q, q.stride(0), (S * q.stride(0), dh), dk, dk.stride(0),
             (Skv * dk.stride(0), dh), B, h)
    out = dres if dres is not None else _new(x, M, H)
    beta = 1.0 if dres is not None else 0.0
    if kv_src is None:
        wq = (sa.query.weight, sa.key.weight, sa.value.weight)
        _linear_bwd_params(A, dqkv, x, wq, (sa.query.bias, sa.key.bias, sa.value.bias))
        return gemm(GEMM_NN, dqkv, _fw(A, wq), out, beta=beta), None           # + gradient of the residual
    _linear_bwd_params(A, dq, x, sa.query.weight, sa.query.bias)
    wkv = (sa.key.weight, sa.value.weight)
    _linear_bwd_params(A, dkv, kv_src, wkv, (sa.key.bias, sa.value.bias))
    dx = gemm(GEMM_NN, dq, _fw(A, sa.query.weight), out, beta=beta)
    dsrc = gemm(GEMM_NN, dkv, _fw(A, wkv), _new(x, kv_src.shape[0], H)) if need_dkv_src else None
    return dx, dsrc


def _dense_norm_fwd(A: ParamArena, mod, h, res, d, save: bool):
    """BertSelfOutput / BertOutput (:561-565, :532-536): LayerNorm(dropout(dense(h)) + res)."""
    o = gemm(GEMM_NT, h, _fw(A, mod.dense.weight), _new(h, h.shape[0], mod.dense.weight.shape[0]), bias=mod.dense.bias)
    seed_h = A.next_seed() if d.p_hidden > 0 else 0
    y, xhat, rstd = ln_fwd(o, res, mod.LayerNorm.weight, mod.LayerNorm.bias, eps=d.eps, p_drop=d.p_hidden, seed=seed_h,
                           save=save)
    return y, ((xhat, rstd, seed_h) if save else None)


def _dense_norm_bwd(A: ParamArena, mod, h, d, saved, dy):
    """Returns (do, dres): gradients of the dense output (dropout mask applied) and of the residual input."""
    xhat, rstd, seed_h = saved
    ln = mod.LayerNorm
    colsum(dy, A.g(ln.weight), b=xhat, accumulate=A.grad_beta(ln.weight) > 0)
    colsum(dy, A.g(ln.bias), accumulate=A.grad_beta(ln.bias) > 0)
    dres, do = ln_bwd(dy, xhat, rstd, ln.weight, p_drop=d.p_hidden, seed=seed_h)
    _linear_bwd_params(A, do, h, mod.dense.weight, mod.dense.bias)
    return do, dres


def _attn_fwd(A: ParamArena, att, x, kv_src, add_mask, d, Skv: int, save: bool):
    """BertAttention / BertCrossAttention (:451-454, :633-636).  Returns (y, saved)."""
    ctx, s_core = _attn_core_fwd(A, att.self, x, kv_src, add_mask, d, Skv, save)
    y, s_out = _dense_norm_fwd(A, att.output, ctx, x, d, save)
    return y, ((ctx, s_core, s_out) if save else None)


def _attn_bwd(A: ParamArena, att, x, kv_src, d, Skv: int, saved, dy, need_dkv_src: bool):
    """Returns (dx, dkv_src)."""
    ctx, s_core, s_out = saved
    dao, dres = _dense_norm_bwd(A, att.output, ctx, d, s_out, dy)
    # without dropout ln_bwd hands back ONE buffer as both gradients: accumulating dx into it below is safe, every
    # reader of dao (weight gradient, dctx) is stream-ordered before that GEMM
    dctx = gemm(GEMM_NN, dao, _fw(A, att.output.dense.weight), _new(x, x.shape[0], x.shape[1]))
    return _attn_core_bwd(A, att.self, x, kv_src, d, Skv, s_core, dctx, dres, need_dkv_src)


def _ffn_fwd(A: ParamArena, layer, x, d, save: bool):
    """BertIntermediate + BertOutput (:548-551, :532-536)."""
    inter = layer.intermediate
    z = gemm(GEMM_NT, x, _fw(A, inter.dense.weight), _new(x, x.shape[0], inter.dense.weight.shape[0]), bias=inter.dense.bias)
    g = act_fwd(z, ACT_GELU)
    y, s_out = _dense_norm_fwd(A, layer.output, g, x, d, save)
    return y, ((z, g, s_out) if save else None)


def _ffn_bwd(A: ParamArena, layer, x, d, saved, dy):
    inter = layer.intermediate
    z, g, s_out = saved
    dfo, dres = _dense_norm_bwd(A, layer.output, g, d, s_out, dy)
    dg = gemm(GEMM_NN, dfo, _fw(A, layer.output.dense.weight), torch.empty_like(z))
    dz = act_bwd(dg, z, ACT_GELU)
    _linear_bwd_params(A, dz, x, inter.dense.weight, inter.dense.bias)
    return gemm(GEMM_NN, dz, _fw(A, inter.dense.weight), dres, beta=1.0)


# =============================================================================================== Functions
class EmbeddingsFn(torch.autograd.Function):
    """BertEmbeddings.forward (Cross_Modal_Interaction_Module.py:398-412)."""

    @staticmethod
    def forward(ctx, anchor, mod, A: ParamArena, ids, tt, d):
        B, S = ids.shape
        H = d.H
        lib = _lib.load()
        y = torch.empty(B * S, H, dtype=F32, device=ids.device)
        xhat = torch.empty_like(y)
        rstd = torch.empty(B * S, dtype=F32, device=ids.device)
        check(lib.icka_x_embed_fwd(ids.data_ptr(), None if tt is None else tt.data_ptr(),
                                   mod.word_embeddings.weight.data_ptr(), mod.position_embeddings.weight.data_ptr(),
                                   mod.token_type_embeddings.weight.data_ptr(), mod.LayerNorm.weight.data_ptr(),
                                   mod.LayerNorm.bias.data_ptr(), y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(), B, S, H,
                                   d.eps, _stream()), "icka_x_embed_fwd")
        seed = A.next_seed() if d.p_hidden > 0 else 0
        if d.p_hidden > 0:
            y = dropout(y, d.p_hidden, seed)
        ctx.mod, ctx.A, ctx.d, ctx.seed = mod, A, d, seed
        ctx.save_for_backward(ids, tt, xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        mod, A, d = ctx.mod, ctx.A, ctx.d
        ids, tt, xhat, rstd = ctx.saved_tensors
        dy = _cf(dy)
        if d.p_hidden > 0:
            dy = dropout(dy, d.p_hidden, ctx.seed)
        ln = mod.LayerNorm
        colsum(dy, A.g(ln.weight), b=xhat, accumulate=A.grad_beta(ln.weight) > 0)
        colsum(dy, A.g(ln.bias), accumulate=A.grad_beta(ln.bias) > 0)
        dpre, _ = ln_bwd(dy, xhat, rstd, ln.weight)
        tables = (mod.word_embeddings.weight, mod.position_embeddings.weight, mod.token_type_embeddings.weight)
        from . import kernels as K
        for t in tables:     # atomically accumulated: a fresh gradient starts from zero
            if A.grad_beta(t) == 0.0:
                K.zero_(A.g(t).view(-1))
        B, S = ids.shape
        pad = mod.word_embeddings.padding_idx
        check(_lib.load().icka_x_embed_scatter(dpre.data_ptr(), ids.data_ptr(), None if tt is None else tt.data_ptr(),
                                               A.g(tables[0]).data_ptr(), A.g(tables[1]).data_ptr(),
                                               A.g(tables[2]).data_ptr(), B, S, d.H, -1 if pad is None else pad,
                                               _stream()), "icka_x_embed_scatter")
        A.flush_final()
        return None, None, None, None, None, None


class BertLayerFn(torch.autograd.Function):
    """BertLayer.forward (:438-442)."""

    @staticmethod
    def forward(ctx, anchor, x, layer, A: ParamArena, add_mask, d):
        save = any(ctx.needs_input_grad)
        x1, s_att = _attn_fwd(A, layer.attention, x, None, add_mask, d, d.S, save)
        x2, s_ffn = _ffn_fwd(A, layer, x1, d, save)
        ctx.layer, ctx.A, ctx.d, ctx.s_att, ctx.s_ffn = layer, A, d, s_att, s_ffn
        ctx.save_for_backward(x, x1)
        return x2

    @staticmethod
    def backward(ctx, dy):
        x, x1 = ctx.saved_tensors
        layer, A, d = ctx.layer, ctx.A, ctx.d
        dx1 = _ffn_bwd(A, layer, x1, d, ctx.s_ffn, _cf(dy))
        dx, _ = _attn_bwd(A, layer.attention, x, None, d, d.S, ctx.s_att, dx1, False)
        if not ctx.A.keep_saved:
            ctx.s_att = ctx.s_ffn = None
        A.flush_final()
        return None, dx, None, None, None, None


class CrossLayerFn(torch.autograd.Function):
    """BertCrossAttentionLayer.forward (:646-650): Q from s1 (text), K/V from s2 (regions), residual = s1."""

    @staticmethod
    def forward(ctx, anchor, s1, s2, layer, A: ParamArena, add_mask, d):
        save = any(ctx.needs_input_grad)
        x1, s_att = _attn_fwd(A, layer.attention, s1, s2, add_mask, d, d.R, save)
        x2, s_ffn = _ffn_fwd(A, layer, x1, d, save)
        ctx.layer, ctx.A, ctx.d, ctx.s_att, ctx.s_ffn = layer, A, d, s_att, s_ffn
        ctx.need_s2 = s2.requires_grad
        ctx.save_for_backward(s1, s2, x1)
        return x2

    @staticmethod
    def backward(ctx, dy):
        s1, s2, x1 = ctx.saved_tensors
        layer, A, d = ctx.layer, ctx.A, ctx.d
        dx1 = _ffn_bwd(A, layer, x1, d, ctx.s_ffn, _cf(dy))
        ds1, ds2 = _attn_bwd(A, layer.attention, s1, s2, d, d.R, ctx.s_att, dx1, ctx.need_s2)
        if not ctx.A.keep_saved:
            ctx.s_att = ctx.s_ffn = None
        A.flush_final()
        return None, ds1, ds2, None, None, None, None


class AttnCoreFn(torch.autograd.Function):
    """BertSelfAttention.forward / BertCoAttention.forward called on their own (:478-506, :590-624)."""

    @staticmethod
    def forward(ctx, anchor, x, kv_src, sa, A: ParamArena, add_mask, d, Skv: int):
        save = any(ctx.needs_input_grad)
        c, saved = _attn_core_fwd(A, sa, x, kv_src, add_mask, d, Skv, save)
        ctx.sa, ctx.A, ctx.d, ctx.Skv, ctx.saved = sa, A, d, Skv, saved
        ctx.need_kv = kv_src is not None and kv_src.requires_grad
        ctx.save_for_backward(x, kv_src)
        return c

    @staticmethod
    def backward(ctx, dc):
        x, kv_src = ctx.saved_tensors
        dx, dsrc = _attn_core_bwd(ctx.A, ctx.sa, x, kv_src, ctx.d, ctx.Skv, ctx.saved, _cf(dc), None, ctx.need_kv)
        if not ctx.A.keep_saved:
            ctx.saved = None
        ctx.A.flush_final()
        return None, dx, dsrc, None, None, None, None, None


class DenseResidualNormFn(torch.autograd.Function):
    """BertSelfOutput.forward / BertOutput.forward (hidden_states, input_tensor) (:561-565, :532-536)."""

    @staticmethod
    def forward(ctx, anchor, h, res, mod, A: ParamArena, d):
        save = any(ctx.needs_input_grad)
        y, saved = _dense_norm_fwd(A, mod, h, res, d, save)
        ctx.mod, ctx.A, ctx.d, ctx.saved = mod, A, d, saved
        ctx.save_for_backward(h)
        return y

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        mod, A = ctx.mod, ctx.A
        do, dres = _dense_norm_bwd(A, mod, h, ctx.d, ctx.saved, _cf(dy))
        dh = gemm(GEMM_NN, do, _fw(A, mod.dense.weight), torch.empty_like(h))
        if not ctx.A.keep_saved:
            ctx.saved = None
        A.flush_final()
        return None, dh, dres, None, None, None


class IntermediateFn(torch.autograd.Function):
    """BertIntermediate.forward (:548-551): gelu(dense(hidden_states))."""

    @staticmethod
    def forward(ctx, anchor, x, inter, A: ParamArena):
        z = gemm(GEMM_NT, x, _fw(A, inter.dense.weight), _new(x, x.shape[0], inter.dense.weight.shape[0]),
                 bias=inter.dense.bias)
        ctx.inter, ctx.A = inter, A
        ctx.save_for_backward(x, z)
        return act_fwd(z, ACT_GELU)

    @staticmethod
    def backward(ctx, dg):
        x, z = ctx.saved_tensors
        inter, A = ctx.inter, ctx.A
        dz = act_bwd(dg, z, ACT_GELU)
        _linear_bwd_params(A, dz, x, inter.dense.weight, inter.dense.bias)
        dx = gemm(GEMM_NN, dz, _fw(A, inter.dense.weight), torch.empty_like(x))
        A.flush_final()
        return None, dx, None, None


class LinearFn(torch.autograd.Function):
    """nn.Linear on a [rows, in] f32 matrix, optional tanh (vismap2text :958, BertPooler :675-681, generic dense)."""

    @staticmethod
    def forward(ctx, anchor, x, lin, A: ParamArena, tanh: bool):
        y = gemm(GEMM_NT, x, _fw(A, lin.weight), _new(x, x.shape[0], lin.weight.shape[0]), bias=lin.bias)
        if tanh:
            act_fwd(y, ACT_TANH, out=y)
        ctx.lin, ctx.A, ctx.tanh = lin, A, tanh
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x, y if tanh else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        lin, A = ctx.lin, ctx.A
        dy = _cf(dy)
        if ctx.tanh:
            dy = act_bwd(dy, y, ACT_TANH)
        _linear_bwd_params(A, dy, x, lin.weight, lin.bias)
        dx = gemm(GEMM_NN, dy, _fw(A, lin.weight), torch.empty(x.shape, dtype=F32, device=x.device)) if ctx.need_dx else None
        A.flush_final()
        return None, dx, None, None, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on the encoder output (:953)."""

    @staticmethod
    def forward(ctx, x, A: ParamArena, p: float):
        ctx.p, ctx.seed = p, A.next_seed()
        return dropout(x, p, ctx.seed)

    @staticmethod
    def backward(ctx, dy):
        return dropout(dy, ctx.p, ctx.seed), None, None


class LayerNormFn(torch.autograd.Function):
    """LayerNorm(x [+ r]) on [rows, H] (BertLayerNorm :509-522; cls_layer_both.forward :879-884)."""

    @staticmethod
    def forward(ctx, anchor, x, r, norm, A: ParamArena, eps: float):
        y, xhat, rstd = ln_fwd(x, r, norm.weight, norm.bias, eps=eps)
        ctx.norm, ctx.A, ctx.two = norm, A, r is not None
        ctx.save_for_backward(xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd = ctx.saved_tensors
        norm, A = ctx.norm, ctx.A
        dy = _cf(dy)
        colsum(dy, A.g(norm.weight), b=xhat, accumulate=A.grad_beta(norm.weight) > 0)
        colsum(dy, A.g(norm.bias), accumulate=A.grad_beta(norm.bias) > 0)
        dpre, _ = ln_bwd(dy, xhat, rstd, norm.weight)
        A.flush_final()
        return None, dpre, (dpre if ctx.two else None), None, None, None


class GatedHeadFn(torch.autograd.Function):
    """my_bert/cl_modeling.py:1363-1371: Gate = sigmoid(Gate_text(seq) + Gate_image(cross));
    logits = classifier(cat(seq, Gate * cross))."""

    @staticmethod
    def forward(ctx, anchor, seq, cross, head, A: ParamArena):
        M, H = seq.shape
        u = gemm(GEMM_NT, seq, _fw(A, head.Gate_text.weight), _new(seq, M, H), bias=head.Gate_text.bias)
        gemm(GEMM_NT, cross, _fw(A, head.Gate_image.weight), u, bias=head.Gate_image.bias, beta=1.0)
        gate = torch.empty_like(u)
        gated = act_fwd(u, ACT_GATE, aux=cross, y2=gate)
        cat = concat2(seq, gated)
        logits = gemm(GEMM_NT, cat, _fw(A, head.classifier.weight), _new(seq, M, head.classifier.weight.shape[0]),
                      bias=head.classifier.bias)
        ctx.head, ctx.A = head, A
        ctx.save_for_backward(seq, cross, gate, cat)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        seq, cross, gate, cat = ctx.saved_tensors
        head, A = ctx.head, ctx.A
        M, H = seq.shape
        dl = _cf(dlogits)
        cls = head.classifier
        _linear_bwd_params(A, dl, cat, cls.weight, cls.bias)
        Wc = _fw(A, cls.weight)
        dseq = gemm(GEMM_NN, dl, Wc[:, :H], _new(seq, M, H))
        dgated = gemm(GEMM_NN, dl, Wc[:, H:], _new(seq, M, H))
        dcross = torch.empty_like(dgated)
        du = act_bwd(dgated, gate, ACT_GATE, aux=cross, dx2=dcross)
        _linear_bwd_params(A, du, seq, head.Gate_text.weight, head.Gate_text.bias)
        _linear_bwd_params(A, du, cross, head.Gate_image.weight, head.Gate_image.bias)
        gemm(GEMM_NN, du, _fw(A, head.Gate_text.weight), dseq, beta=1.0)
        gemm(GEMM_NN, du, _fw(A, head.Gate_image.weight), dcross, beta=1.0)
        A.flush_final()
        return None, dseq, dcross, None, None


class SampleGateFn(torch.autograd.Function):
    """mode 0: out = g*a + (1-g)*c, g = sigmoid(gate[b]) (Cross_Modal_Interaction_Module.py:1035-1036);
    mode 1: out = softmax(gate[b])[1] * a (gate_cl_modeling.py:1369-1373)."""

    @staticmethod
    def forward(ctx, a, c, gate, mode: int, B: int, S: int):
        _mat(a, "a")
        H = a.shape[1]
        out = _new(a, a.shape[0], H)
        check(_lib.load().icka_x_sample_gate_fwd(a.data_ptr(), a.stride(0), None if c is None else c.data_ptr(),
                                                 0 if c is None else c.stride(0), gate.data_ptr(), mode, out.data_ptr(), H,
                                                 B, S, H, _stream()), "icka_x_sample_gate_fwd")
        ctx.mode, ctx.B, ctx.S = mode, B, S
        ctx.save_for_backward(a, c, gate)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, c, gate = ctx.saved_tensors
        dout = _cf(dout)
        H = a.shape[1]
        da = torch.empty_like(dout)
        dc = torch.empty_like(dout) if c is not None else None
        dgate = torch.empty_like(gate)
        check(_lib.load().icka_x_sample_gate_bwd(dout.data_ptr(), H, a.data_ptr(), a.stride(0),
                                                 None if c is None else c.data_ptr(), 0 if c is None else c.stride(0),
                                                 gate.data_ptr(), ctx.mode, da.data_ptr(), H,
                                                 None if dc is None else dc.data_ptr(), H, dgate.data_ptr(), ctx.B, ctx.S, H,
                                                 _stream()), "icka_x_sample_gate_bwd")
        return da, dc, dgate, None, None, None


class CrsFn(torch.autograd.Function):
    """crs_classifier(cat(seq, cross).view(B, -1)) of gate_cl (gate_cl_modeling.py:1364-1366) -> f32 [B,2]."""

    @staticmethod
    def forward(ctx, anchor, seq, cross, lin, A: ParamArena, B: int, S: int):
        H = seq.shape[1]
        cat = concat2(seq, cross).view(B, S * 2 * H)
        crs = gemm(GEMM_NT, cat, _fw(A, lin.weight), _new(seq, B, lin.weight.shape[0]), bias=lin.bias)
        ctx.lin, ctx.A, ctx.H = lin, A, H
        ctx.save_for_backward(cat)
        return crs

    @staticmethod
    def backward(ctx, dcrs):
        (cat,) = ctx.saved_tensors
        lin, A, H = ctx.lin, ctx.A, ctx.H
        dcrs = _cf(dcrs)
        _linear_bwd_params(A, dcrs, cat, lin.weight, lin.bias)
        dcat = gemm(GEMM_NN, dcrs, _fw(A, lin.weight), torch.empty_like(cat)).view(-1, 2 * H)
        A.flush_final()
        return None, dcat[:, :H], dcat[:, H:], None, None, None, None


class TokenCEFn(torch.autograd.Function):
    """Token-level cross-entropy, mean over valid tokens, f32 logit gradient."""

    @staticmethod
    def forward(ctx, logits, labels, mask):
        M, Cn = logits.shape
        lib = _lib.load()
        stats = torch.zeros(3, dtype=F32, device=logits.device)
        dl = torch.empty(M, Cn, dtype=F32, device=logits.device)
        check(lib.icka_x_token_ce(logits.data_ptr(), logits.stride(0), labels.data_ptr(), mask.data_ptr(),
                                  stats[0:1].data_ptr(), stats[1:2].data_ptr(), dl.data_ptr(), M, Cn, _stream()),
              "icka_x_token_ce")
        check(lib.icka_scalar_ratio(stats[2:3].data_ptr(), stats[0:1].data_ptr(), stats[1:2].data_ptr(), _stream()),
              "icka_scalar_ratio")
        ctx.save_for_backward(dl, stats)
        return stats[2:3].view(())

    @staticmethod
    def backward(ctx, dloss):
        dl, stats = ctx.saved_tensors
        g = dloss.reshape(1).contiguous()
        if g.dtype != F32:
            raise TypeError("loss gradient must be f32")
        out = torch.empty_like(dl)
        check(_lib.load().icka_x_scale_by_ratio(dl.data_ptr(), out.data_ptr(), g.data_ptr(), stats[1:2].data_ptr(),
                                                dl.numel(), _stream()), "icka_x_scale_by_ratio")
        return out, None, None


def regions_to_tokens(v: torch.Tensor, B: int, R: int, layout: int) -> torch.Tensor:
    _f32(v, "visual_embeds_att")
    v = v if v.is_contiguous() else v.contiguous()
    out = torch.empty(B * R, 2048, dtype=F32, device=v.device)
    check(_lib.load().icka_x_regions_to_tokens(v.data_ptr(), out.data_ptr(), B, R, 2048, layout, _stream()),
          "icka_x_regions_to_tokens")
    return out


# =============================================================================================== section 8(f) rows
class FanOutFn(torch.autograd.Function):
    """y1 = y2 = ... = x for a tensor with several consumers; the backward sums the branch gradients with icka_x_add."""

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
            g = _cf(g)
            if acc is None:
                acc = g
            else:
                g2, a2 = g.view(-1, g.shape[-1]), acc.view(-1, g.shape[-1])
                out = torch.empty_like(a2)
                check(_lib.load().icka_x_add(a2.data_ptr(), a2.stride(0), g2.data_ptr(), g2.stride(0), out.data_ptr(),
                                             out.stride(0), a2.shape[0], a2.shape[1], _stream()), "icka_x_add")
                acc = out.view(g.shape)
        return acc, None


class LstmFn(torch.autograd.Function):
    """nn.LSTM(H_in, H, batch_first=True, bidirectional=True) (Cross_Modal_Interaction_Module.py:905-908, call :1042) in
    f32: input projection of all steps and both directions = one GEMM; per step one batched (2 directions) recurrent GEMM
    accumulating into the gate buffer + one pointwise cell launch; backward the same in reverse, then the weight / bias /
    input gradients as GEMMs and column sums over all steps."""

    @staticmethod
    def forward(ctx, anchor, x, mod, A: ParamArena, B: int, S: int):
        H = mod.hidden_size
        M = B * S
        lib = _lib.load()
        wih = (mod.weight_ih_l0, mod.weight_ih_l0_reverse)
        whh = _fw(A, (mod.weight_hh_l0, mod.weight_hh_l0_reverse))            # [2*4H, H]
        bsum = _new(x, 1, 8 * H)
        bi, bh = _fb(A, (mod.bias_ih_l0, mod.bias_ih_l0_reverse)), _fb(A, (mod.bias_hh_l0, mod.bias_hh_l0_reverse))
        check(lib.icka_x_add(bi.data_ptr(), 8 * H, bh.data_ptr(), 8 * H, bsum.data_ptr(), 8 * H, 1, 8 * H, _stream()),
              "icka_x_add")
        gates = gemm(GEMM_NT, x, _fw(A, wih), _new(x, M, 8 * H), bias=bsum.view(-1))
        y, c_all, hprev = _new(x, M, 2 * H), _new(x, M, 2 * H), _new(x, M, 2 * H)
        for k in range(S):
            if k:
                # gates[b, t_dir, dir] += h_{t-1, dir} . W_hh_dir^T ; forward direction t = k (previous k-1), reverse
                # direction t = S-1-k (previous S-k): the two problems differ by constant element offsets
                a0, a1 = (k - 1) * 2 * H, (S - k) * 2 * H + H
                c0, c1 = k * 8 * H, (S - 1 - k) * 8 * H + 4 * H
                gemm_raw(GEMM_NT, B, 4 * H, H, y.view(-1)[a0:], S * 2 * H, (a1 - a0, 0), whh, H, (4 * H * H, 0),
                         gates.view(-1)[c0:], S * 8 * H, (c1 - c0, 0), 2, 1, beta=1.0)
            check(lib.icka_x_lstm_cell_fwd(gates.data_ptr(), c_all.data_ptr(), y.data_ptr(), hprev.data_ptr(), B, S, H, k,
                                           _stream()), "icka_x_lstm_cell_fwd")
        ctx.mod, ctx.A, ctx.dims = mod, A, (B, S, H)
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x, gates, c_all, hprev)
        ctx.mark_non_differentiable(c_all)
        return y, c_all

    @staticmethod
    def backward(ctx, dy, _dc):
        x, act, c_all, hprev = ctx.saved_tensors
        mod, A = ctx.mod, ctx.A
        B, S, H = ctx.dims
        lib = _lib.load()
        dy = _cf(dy)
        whh = _fw(A, (mod.weight_hh_l0, mod.weight_hh_l0_reverse))
        dgates = act                               # the saved activations are consumed in place (one backward per forward)
        dh, dc = _new(x, 2, B, H), _new(x, 2, B, H)
        for k in range(S - 1, -1, -1):
            check(lib.icka_x_lstm_cell_bwd(dy.data_ptr(), dh.data_ptr(), dc.data_ptr(), dgates.data_ptr(), c_all.data_ptr(),
                                           B, S, H, k, int(k == S - 1), _stream()), "icka_x_lstm_cell_bwd")
            if k:
                # dh_rec[dir] = dgates[b, t_dir, dir] . W_hh_dir   (reaches h of step k-1)
                c0, c1 = k * 8 * H, (S - 1 - k) * 8 * H + 4 * H
                gemm_raw(GEMM_NN, B, H, 4 * H, dgates.view(-1)[c0:], S * 8 * H, (c1 - c0, 0), whh, H, (4 * H * H, 0),
                         dh, H, (B * H, 0), 2, 1)
        wih = (mod.weight_ih_l0, mod.weight_ih_l0_reverse)
        gemm(GEMM_TN, dgates, x, A.g_cat(wih), beta=A.grad_beta(wih))
        for d, w in enumerate((mod.weight_hh_l0, mod.weight_hh_l0_reverse)):
            gemm(GEMM_TN, dgates[:, d * 4 * H:(d + 1) * 4 * H], hprev[:, d * H:(d + 1) * H], A.g(w), beta=A.grad_beta(w))
        for bs in ((mod.bias_ih_l0, mod.bias_ih_l0_reverse), (mod.bias_hh_l0, mod.bias_hh_l0_reverse)):
            colsum(dgates, A.g_cat(bs), accumulate=A.grad_beta(bs) > 0)
        dx = gemm(GEMM_NN, dgates, _fw(A, wih), torch.empty_like(x)) if ctx.need_dx else None
        A.flush_final()
        return None, dx, None, None, None, None


class PromptEmbeddingsFn(torch.autograd.Function):
    """Embeddings of the prompt-accepting encoder stage in f32 (see ops.PromptEmbeddingsFn)."""

    @staticmethod
    def forward(ctx, anchor, prompt, mod, A: ParamArena, ids, src, d, pos_offset: int):
        B, S_in = ids.shape
        S, H = src.shape[0], d.H
        prompt = _cf(prompt)
        y = torch.empty(B * S, H, dtype=F32, device=ids.device)
        xhat = torch.empty_like(y)
        rstd = torch.empty(B * S, dtype=F32, device=ids.device)
        check(_lib.load().icka_x_embed_prompt_fwd(ids.data_ptr(), src.data_ptr(), prompt.data_ptr(),
                                                  mod.word_embeddings.weight.data_ptr(),
                                                  mod.position_embeddings.weight.data_ptr(),
                                                  mod.token_type_embeddings.weight.data_ptr(), mod.LayerNorm.weight.data_ptr(),
                                                  mod.LayerNorm.bias.data_ptr(), y.data_ptr(), xhat.data_ptr(), rstd.data_ptr(),
                                                  B, S_in, S, prompt.shape[1], H, pos_offset, d.eps, _stream()),
              "icka_x_embed_prompt_fwd")
        seed = A.next_seed() if d.p_hidden > 0 else 0
        if d.p_hidden > 0:
            y = dropout(y, d.p_hidden, seed)
        ctx.mod, ctx.A, ctx.d, ctx.seed, ctx.pos_offset, ctx.pshape = mod, A, d, seed, pos_offset, tuple(prompt.shape)
        ctx.save_for_backward(ids, src, xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import kernels as K
        mod, A, d = ctx.mod, ctx.A, ctx.d
        ids, src, xhat, rstd = ctx.saved_tensors
        dy = _cf(dy)
        if d.p_hidden > 0:
            dy = dropout(dy, d.p_hidden, ctx.seed)
        ln = mod.LayerNorm
        colsum(dy, A.g(ln.weight), b=xhat, accumulate=A.grad_beta(ln.weight) > 0)
        colsum(dy, A.g(ln.bias), accumulate=A.grad_beta(ln.bias) > 0)
        dpre, _ = ln_bwd(dy, xhat, rstd, ln.weight)
        tables = (mod.word_embeddings.weight, mod.position_embeddings.weight, mod.token_type_embeddings.weight)
        for t in tables:
            if A.grad_beta(t) == 0.0:
                K.zero_(A.g(t).view(-1))
        dprompt = torch.empty(ctx.pshape, dtype=F32, device=dy.device)
        pad = mod.word_embeddings.padding_idx
        B, S_in = ids.shape
        check(_lib.load().icka_x_embed_prompt_scatter(dpre.data_ptr(), ids.data_ptr(), src.data_ptr(),
                                                      A.g(tables[0]).data_ptr(), A.g(tables[1]).data_ptr(),
                                                      A.g(tables[2]).data_ptr(), dprompt.data_ptr(), B, S_in, src.shape[0],
                                                      ctx.pshape[1], d.H, ctx.pos_offset, -1 if pad is None else pad, _stream()),
              "icka_x_embed_prompt_scatter")
        A.flush_final()
        return None, dprompt, None, None, None, None, None, None


def prompt_mapping(A: ParamArena, x, lin1, lin2, p: float):
    """nn.Sequential(Dropout(p), Linear, Tanh(), Dropout(p), Linear) (Cross_Modal_Interaction_Module.py:914-928)."""
    if p > 0:
        x = DropoutFn.apply(x, A, p)
    h = LinearFn.apply(A.anchor, x, lin1, A, True)
    if p > 0:
        h = DropoutFn.apply(h, A, p)
    return LinearFn.apply(A.anchor, h, lin2, A, False)
