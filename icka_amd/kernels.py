"""Tensor-level launchers over the C-ABI (include/icka_hip.h).

PyTorch is used here only for device memory and the current HIP stream: each function checks device / dtype /
contiguity on the host, then hands raw device pointers + sizes + the stream to libicka_hip.so.  Tensors must live
on a ROCm device; there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_ADD, EPI_ADD_RELU, EPI_DGELU, EPI_GATE, EPI_GELU, EPI_NONE, EPI_RELU, EPI_TANH, GEMM_NN, GEMM_NT, GEMM_TN,
                   GemmDesc, check)

BF16 = torch.bfloat16
F32 = torch.float32
F16 = torch.float16


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CUR_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """Raw handle of torch's current HIP stream.  (torch.cuda.current_stream().cuda_stream builds a Stream object per call:
    ~10 us of host time, i.e. ~2 ms of an eager c2 step's ~230 launches -- tools/eager_profile.py.)"""
    if _RAW_STREAM is not None and _CUR_DEVICE is not None:
        return _RAW_STREAM(_CUR_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def _dev(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise TypeError("%s must be a ROCm device tensor (icka_amd has no CPU path), got %s" % (name, t.device))


def _mat(t: torch.Tensor, name: str, dtype=BF16) -> None:
    # one-expression fast path (this runs ~700 times per eager c2 step); the slow path below words the error
    if t.is_cuda and t.dtype is dtype and t.dim() == 2 and (t.shape[1] <= 1 or t.stride(1) == 1):
        return
    _dev(t, name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be a 2-D row-major view (stride(1)==1), got shape %s strides %s"
                         % (name, tuple(t.shape), t.stride()))


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _ld(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.stride(0)


# ------------------------------------------------------------------------------------------------------- GEMM
_PROF = None   # None, or the list of GEMM launches recorded while profile_gemm(True) is active


def profile_gemm(enable: bool, reps: int = 5):
    """bench.py roofline leg.  profile_gemm(True): start recording every GEMM launch (descriptor + operand tensors,
    kept alive).  profile_gemm(False): re-issue the recorded launches back to back, ``reps`` times, between ONE pair
    of HIP events on the launch stream (so no host gap or per-launch event cost is inside the bracket) and return
    (algorithmic FLOPs, elapsed ms, launches, algorithmic bytes) of what ran inside the bracket."""
    global _PROF
    if enable:
        _PROF = []
        return None
    rec, _PROF = _PROF or [], None
    lib = _lib.load()
    global last_fused_profile
    last_fused_profile = None
    if not rec:
        return 0.0, 0.0, 0, 0.0
    plain = [r for r in rec if r[0] < 2]
    fused = [r for r in rec if r[0] == 2]      # icka_gemm_ln launches: a GEMM with its LayerNorm phase in the same kernel
    fused_qa = [r for r in rec if r[0] == 3]   # icka_gemm_qkv_attn launches: the QKV projection with its attention in the same kernel

    def replay(which):
        for kind, payload, n, _keep in which:
            if kind == 0:
                check(lib.icka_gemm(C.byref(payload), _stream()), "icka_gemm")
            elif kind == 1:
                check(lib.icka_gemm_grouped(payload, n, _stream()), "icka_gemm_grouped")
            elif kind == 2:
                check(lib.icka_gemm_ln(*payload), "icka_gemm_ln")
            else:
                check(lib.icka_gemm_qkv_attn(*payload), "icka_gemm_qkv_attn")

    def timed(which):
        replay(which)   # warm
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            replay(which)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    fl = lambda which: reps * sum(2.0 * d.M * d.N * d.K for _, _, _, ds in which for d in ds)
    if fused:
        # the fused launches are timed in a bracket of their own: their duration contains the HBM-bound LayerNorm phase, so they
        # are reported beside the MFMA-bound class, not inside it
        last_fused_profile = {"flops": fl(fused), "ms": timed(fused), "launches": reps * len(fused)}
    global last_fused_qkv_profile
    last_fused_qkv_profile = None
    if fused_qa:    # likewise: the launch contains the whole-head attention of its tiles (GEMM FLOPs only are counted)
        last_fused_qkv_profile = {"flops": fl(fused_qa), "ms": timed(fused_qa), "launches": reps * len(fused_qa)}
    ms = timed(plain) if plain else 0.0
    nbytes = reps * sum(_gemm_bytes(d) for _, _, _, ds in plain for d in ds)
    return fl(plain), ms, reps * len(plain), nbytes


last_fused_qkv_profile = None   # the same for the recorded icka_gemm_qkv_attn launches
last_fused_profile = None     # profile_gemm(False): {"flops", "ms", "launches"} of the recorded icka_gemm_ln launches, or None


def _gemm_bytes(d) -> float:
    """algorithmic HBM bytes of one GEMM: both operands once + the output once (bf16 in, bf16/f32 out)"""
    return 2.0 * (d.M * d.K + d.N * d.K) + d.M * d.N * (4.0 if d.c_is_f32 == 1 else 2.0) + (2.0 * d.M * d.N if d.C3 else 0.0)


def gemm_desc(op: int, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, *, bias: Optional[torch.Tensor] = None,
              epilogue: int = EPI_NONE, aux: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None,
              alpha: float = 1.0, beta: float = 0.0, A2: Optional[torch.Tensor] = None,
              B2: Optional[torch.Tensor] = None, bias2: Optional[torch.Tensor] = None,
              colsum_out: Optional[torch.Tensor] = None, colsum_accumulate: bool = False,
              out3: Optional[torch.Tensor] = None, out3_only: bool = False, tune: int = 0) -> GemmDesc:
    """Validate the operands and fill an ``icka_gemm_desc`` (the tensors must stay alive until it is launched).
    Operands are bf16, or -- op NT only, the "mixed16" forward GEMMs -- both fp16; ``out`` may then be fp16 too, with
    ``out3`` an optional bf16 copy of it.  With an f32 ``out``, ``out3`` is the data-parallel wire copy (bf16 of the final,
    beta-accumulated value) the weight-gradient GEMMs write for dp.GradReducer: op TN only.  ``tune``: per-call overrides of
    the launch heuristics (``gemm_tune(...)``; 0 = the library's choice) -- results do not depend on it."""
    odt = A.dtype if A.dtype == F16 else BF16     # operand dtype of this launch
    if odt == F16 and op != GEMM_NT:
        raise ValueError("fp16 operands: NT (forward) GEMMs only")
    _mat(A, "A", odt); _mat(B, "B", odt)
    if op == GEMM_NT:
        M, K = A.shape; N, Kb = B.shape
    elif op == GEMM_NN:
        M, K = A.shape; Kb, N = B.shape
    elif op == GEMM_TN:
        K, M = A.shape; Kb, N = B.shape
    else:
        raise ValueError("bad op")
    if K != Kb:
        raise ValueError("reduction mismatch: %s vs %s" % (tuple(A.shape), tuple(B.shape)))
    K1 = 0
    if A2 is not None or B2 is not None:
        if A2 is None or B2 is None:
            raise ValueError("A2 and B2 go together")
        _mat(A2, "A2", odt); _mat(B2, "B2", odt)
        K2 = A2.shape[0] if op == GEMM_TN else A2.shape[1]
        K1, K = K, K + K2
    _dev(out, "out")
    if out.dtype not in (BF16, F32, F16) or out.dim() != 2 or tuple(out.shape) != (M, N) or (N > 1 and out.stride(1) != 1):
        raise ValueError("out must be [%d,%d] bf16/f32/fp16 row-major, got %s %s" % (M, N, tuple(out.shape), out.dtype))
    if out.dtype == F16 and (beta != 0.0 or colsum_out is not None):
        raise ValueError("fp16 outputs are plain forward outputs (no accumulate, no fused column sums)")
    if out3 is not None:
        if out.dtype not in (F16, F32):
            raise ValueError("out3 is the bf16 copy of an fp16 main output, or the data-parallel wire copy of an f32 one")
        _mat(out3, "out3")
        if tuple(out3.shape) != (M, N):
            raise ValueError("out3 must be [%d,%d]" % (M, N))
        if out.dtype == F32 and op != GEMM_TN:
            raise ValueError("out3 beside an f32 output (the data-parallel wire copy) exists on weight-gradient (TN) GEMMs only")
    d = GemmDesc()
    d.op, d.M, d.N, d.K, d.K1 = op, M, N, K, K1
    d.A, d.lda, d.B, d.ldb = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0)
    if A2 is not None:       # (ctypes zero-initialises the struct: optional fields are only written when present)
        d.A2, d.lda2, d.B2, d.ldb2 = A2.data_ptr(), A2.stride(0), B2.data_ptr(), B2.stride(0)
    d.C, d.ldc, d.c_is_f32 = out.data_ptr(), out.stride(0), {BF16: 0, F32: 1, F16: 2}[out.dtype]
    if odt == F16:
        d.ab_f16 = 1
    if out3 is not None:
        d.C3, d.ldc3 = out3.data_ptr(), out3.stride(0)
    if out3_only:
        if not (out3 is not None and out.dtype == F32 and beta == 0.0 and epilogue == EPI_NONE):
            raise ValueError("out3_only: an f32 output with a wire copy, beta == 0, no epilogue")
        d.c3_only = 1
    if out2 is not None:
        _mat(out2, "out2")
        d.C2, d.ldc2 = out2.data_ptr(), out2.stride(0)
    if aux is not None:
        _mat(aux, "aux", F16 if aux.dtype == F16 else BF16)
        d.aux, d.ldaux = aux.data_ptr(), aux.stride(0)
        if aux.dtype == F16:
            d.aux_f16 = 1
    if bias is not None:
        if not bias.is_cuda or bias.dtype != F32 or bias.numel() != N or not bias.is_contiguous():
            _dev(bias, "bias")
            raise ValueError("bias must be contiguous f32 [N]")
        d.bias = bias.data_ptr()
    if bias2 is not None:
        if bias2.dtype != F32 or bias2.numel() != N or not bias2.is_contiguous() or not bias2.is_cuda:
            raise ValueError("bias2 must be contiguous device f32 [N]")
        d.bias2 = bias2.data_ptr()
    d.alpha, d.beta, d.epilogue = alpha, beta, epilogue
    if tune:
        d.tune = int(tune)
    if colsum_out is not None:
        if op != GEMM_TN or colsum_out.dtype != F32 or colsum_out.numel() != M or not colsum_out.is_contiguous():
            raise ValueError("colsum_out: contiguous f32 [M] with op TN")
        if M % 128 or N % 128 or K % 64:
            raise ValueError("fused column sums need the aligned fast path (M, N % 128 == 0, K % 64 == 0)")
        d.colsum_out, d.colsum_accumulate = colsum_out.data_ptr(), int(colsum_accumulate)
    if _PROF is not None:   # the recorded launch is re-issued later: its operands must outlive the step
        d._keep = (A, B, out, bias, aux, out2, A2, B2, bias2, colsum_out, out3)
    return d


def gemm_tune(*, ring: Optional[int] = None, tile_n: Optional[int] = None, wide_tiles: Optional[bool] = None,
              direct_epilogue: Optional[bool] = None, warp_specialized: Optional[int] = None, w3_grid: Optional[int] = None,
              big_tiles: Optional[int] = None, ablation: Optional[int] = None) -> int:
    """The ``icka_gemm_desc.tune`` word (include/icka_hip.h, ICKA_TUNE_*): per-call overrides of single launch heuristics, for
    tests of the alternative kernels and for probes.  There is no process-wide setter; the same fields can be given to a whole
    process through the ICKA_TUNE_GEMM_* environment (read once when the library loads)."""
    t = 0
    if ring is not None:
        if ring not in (2, 3, 4, 5):
            raise ValueError("ring: 2 .. 5")
        t |= ring
    if tile_n is not None:
        if tile_n not in (96, 128):
            raise ValueError("tile_n: 96 or 128")
        t |= (1 if tile_n == 96 else 2) << 4
    if wide_tiles is not None:
        t |= (2 if wide_tiles else 1) << 8
    if direct_epilogue is not None:
        t |= (2 if direct_epilogue else 1) << 12
    if warp_specialized is not None:
        if warp_specialized not in (0, 1, 2, 3):
            raise ValueError("warp_specialized: 0 .. 3")
        t |= (warp_specialized + 1) << 16
    if w3_grid is not None:
        if w3_grid not in (1, 2, 4, 8):
            raise ValueError("w3_grid: 1, 2, 4 or 8")
        t |= w3_grid << 20
    if big_tiles is not None:
        if big_tiles not in (0, 1, 2):
            raise ValueError("big_tiles: 0 .. 2")
        t |= (big_tiles + 1) << 24
    if ablation is not None:
        if ablation not in (1, 2, 3):
            raise ValueError("ablation: 1 .. 3 (diagnostic builds)")
        t |= ablation << 28
    return t


def gemm(op: int, A: torch.Tensor, B: torch.Tensor, out: torch.Tensor, **kw) -> torch.Tensor:
    """out[M,N] = epilogue(alpha * op(A,B) [+ op(A2,B2)] + bias) + beta*out.   op: GEMM_NT / GEMM_NN / GEMM_TN."""
    lib = _lib.load()
    d = gemm_desc(op, A, B, out, **kw)
    if _PROF is not None:
        _PROF.append((0, d, 1, [d]))
    check(lib.icka_gemm(C.byref(d), _stream()), "icka_gemm")
    return out


def slab_reduction(partials: torch.Tensor, nslab: int, H: int, outs, accumulate: bool, slab_stride: int = 0,
                   offset: int = 0):
    """Descriptor of  outs[slot][c] (+)= sum_b partials[offset + b*slab_stride + slot*H + c]  (c < H); rides on a
    gemm_grouped launch (``reductions=``).  Default stride: the LayerNorm slab layout left by ln_bwd_slabs."""
    r = _lib.SlabReduction()
    if slab_stride == 0:
        slab_stride = _lib.load().icka_ln_slab_slots() * H
    r.partials, r.slab_stride, r.nslab, r.H = partials.data_ptr() + 4 * offset, slab_stride, nslab, H
    r.nslots, r.accumulate = len(outs), int(bool(accumulate))
    for i, o in enumerate(outs):
        _dev(o, "reduction output")
        if o.dtype != F32 or o.numel() != H or not o.is_contiguous():
            raise ValueError("reduction outputs must be contiguous f32 with %d elements" % H)
        r.out[i] = o.data_ptr()
    r._keep = (partials,) + tuple(outs)
    return r


def cls_head_fwd(seq, gated, W, bias, logits):
    """logits f32 [M,C] = [seq | gated] . W^T + bias (W [C, 2H]); seq, gated and W all bf16, or all fp16 (mixed16)."""
    dt = seq.dtype if seq.dtype == F16 else BF16
    _mat(seq, "seq", dt); _mat(gated, "gated", dt)
    M, H = seq.shape
    Cn = W.shape[0]
    if not (seq.is_contiguous() and gated.is_contiguous() and W.is_contiguous() and logits.is_contiguous()):
        raise ValueError("cls_head_fwd: contiguous operands")
    if W.dtype != dt or tuple(W.shape) != (Cn, 2 * H) or logits.dtype != F32 or tuple(logits.shape) != (M, Cn):
        raise ValueError("cls_head_fwd: W %s [C,2H], logits f32 [M,C]" % dt)
    lib = _lib.load()
    fn = lib.icka_cls_head_fwd_h if dt == F16 else lib.icka_cls_head_fwd
    check(fn(seq.data_ptr(), gated.data_ptr(), W.data_ptr(), _ptr(bias), logits.data_ptr(), M, H, Cn, _stream()),
          "icka_cls_head_fwd")
    return logits


def cls_head_bwd(dl, seq, gated, gate, cross, W, dseq, du, dcross, partials):
    """Returns the number of slabs written to ``partials`` (see icka_cls_head_bwd)."""
    M, H = seq.shape
    Cn = W.shape[0]
    for n, t in (("seq", seq), ("gated", gated), ("gate", gate), ("cross", cross), ("dseq", dseq), ("du", du),
                 ("dcross", dcross)):
        _mat(t, n)
        if not t.is_contiguous() or tuple(t.shape) != (M, H):
            raise ValueError("%s must be contiguous bf16 [M,H]" % n)
    _mat(dl, "dl")
    lib = _lib.load()
    check(lib.icka_cls_head_bwd(dl.data_ptr(), dl.stride(0), seq.data_ptr(), gated.data_ptr(), gate.data_ptr(),
                                cross.data_ptr(), W.data_ptr(), dseq.data_ptr(), du.data_ptr(), dcross.data_ptr(),
                                partials.data_ptr(), M, H, Cn, _stream()), "icka_cls_head_bwd")
    return lib.icka_cls_head_bwd_slabs(M)


def gemm_grouped(descs, reductions=None) -> None:
    """Launch several GEMMs (gemm_desc results) at once; same-layout fast-path problems share one launch.
    ``reductions``: slab_reduction descriptors (at most 4) summed by extra blocks of the same launch."""
    reductions = reductions or []
    if not descs and not reductions:
        return
    lib = _lib.load()
    arr = (GemmDesc * max(len(descs), 1))(*descs)
    if _PROF is not None and descs:
        _PROF.append((1, arr, len(descs), list(descs)))
    if not reductions:
        check(lib.icka_gemm_grouped(arr, len(descs), _stream()), "icka_gemm_grouped")
        return
    for i in range(0, len(reductions), 4):   # the ABI takes 4 per call; GEMMs go with the first chunk
        chunk = reductions[i:i + 4]
        rarr = (_lib.SlabReduction * len(chunk))(*chunk)
        check(lib.icka_gemm_grouped_ex(arr if i == 0 else None, len(descs) if i == 0 else 0, rarr, len(chunk),
                                       _stream()), "icka_gemm_grouped_ex")


# ------------------------------------------------------------------------------------------------- LayerNorm
def _kind(t: torch.Tensor) -> int:
    """element-type code of the LayerNorm launchers: 0 bf16, 1 f32, 2 fp16"""
    return {BF16: 0, F32: 1, F16: 2}[t.dtype]


def ln_fwd(x, bias, residual, gamma, beta, y, *, y2=None, y_f32=None, y_f16=None, xhat=None, rstd=None, eps=1e-12,
           p_drop=0.0, seed=0):
    """x / residual may be bf16, f32 or fp16 [M,H] row-major; y (and y2) bf16; twin copy of the output: y_f32 (contiguous
    f32) or y_f16 (contiguous fp16, the "mixed16" forward operand + residual) -- at most one of the two."""
    lib = _lib.load()
    _mat(x, "x", x.dtype if x.dtype in (BF16, F32, F16) else BF16); _mat(y, "y")
    if residual is not None:
        _mat(residual, "residual", residual.dtype if residual.dtype in (BF16, F32, F16) else BF16)
    if y_f32 is not None and y_f16 is not None:
        raise ValueError("one twin copy: y_f32 or y_f16")
    M, H = x.shape
    twin = y_f16 if y_f16 is not None else y_f32
    if twin is not None and (twin.dtype != (F16 if y_f16 is not None else F32) or not twin.is_contiguous()
                             or tuple(twin.shape) != (M, H)):
        raise ValueError("twin output must be contiguous [M,H] f32 (y_f32) / fp16 (y_f16)")
    fn = lib.icka_ln_fwd_h if y_f16 is not None else lib.icka_ln_fwd
    check(fn(x.data_ptr(), x.stride(0), _kind(x), _ptr(bias), _ptr(residual), _ld(residual),
             0 if residual is None else _kind(residual), gamma.data_ptr(), beta.data_ptr(),
             y.data_ptr(), y.stride(0), _ptr(y2), _ld(y2), _ptr(twin), _ptr(xhat), _ptr(rstd),
             M, H, eps, p_drop, seed, _stream()), "icka_ln_fwd")
    return y


class GemmLnHandoffError(RuntimeError):
    """A stripe wait of a fused dense + LayerNorm launch gave up: the rows it finished are NaN."""


_GEMM_LN_ERR = None      # one pinned (host-mapped) error word for the process: polled by the host without a device sync
_GEMM_LN_ERR_NP = None   # numpy view of it (a host read through it costs ~0.1 us: it sits in front of every fused launch)
_GEMM_LN_ERR_PTR = 0


def _gemm_ln_err_word() -> int:
    """Device-visible address of the pinned error word (allocated on first use: never under stream capture -- a model's first
    forward, and every capture's warm-up, run eagerly)."""
    global _GEMM_LN_ERR, _GEMM_LN_ERR_NP, _GEMM_LN_ERR_PTR
    if _GEMM_LN_ERR is None:
        _GEMM_LN_ERR = torch.zeros(16, dtype=torch.int32).pin_memory()
        _GEMM_LN_ERR_NP = _GEMM_LN_ERR.numpy()
        _GEMM_LN_ERR_PTR = _GEMM_LN_ERR.data_ptr()
    return _GEMM_LN_ERR_PTR


def gemm_ln_sync(device) -> torch.Tensor:
    """Zeroed counter words for ``gemm_ln`` (icka_gemm_ln_sync_words; the launches leave them zero: one buffer per stream)."""
    return torch.zeros(_lib.load().icka_gemm_ln_sync_words(), dtype=torch.int32, device=device)


def gemm_ln(h, w, o, bias, residual, gamma, beta, y, sync, *, y_f32=None, y_f16=None, xhat=None, rstd=None, eps=1e-12,
            p_drop=0.0, seed=0) -> bool:
    """dense (h @ w^T -> o, f32) + bias + dropout + residual + LayerNorm -> y (and twin / xhat / rstd) as ONE launch
    (icka_gemm_ln) -- bitwise ``gemm(NT, h, w, o)`` followed by ``ln_fwd(o, bias, residual, gamma, beta, y, ...)``.  Returns False
    (nothing launched) when the shape is not eligible: the caller then makes the two calls.  A stripe wait of an EARLIER fused
    launch that gave up is raised here (gemm_ln_check_error: a host read of a pinned word)."""
    gemm_ln_check_error("detected before the next fused launch")
    lib = _lib.load()
    d = gemm_desc(GEMM_NT, h, w, o)
    _mat(y, "y")
    if residual is not None:
        _mat(residual, "residual", residual.dtype if residual.dtype in (BF16, F32, F16) else BF16)
    twin = y_f16 if y_f16 is not None else y_f32
    M, N = o.shape
    if twin is not None and (twin.dtype != (F16 if y_f16 is not None else F32) or not twin.is_contiguous() or tuple(twin.shape) != (M, N)):
        raise ValueError("twin output must be contiguous [M,N] f32 (y_f32) / fp16 (y_f16)")
    args = (d, _ptr(bias), _ptr(residual), _ld(residual), 0 if residual is None else _kind(residual), gamma.data_ptr(),
            beta.data_ptr(), y.data_ptr(), y.stride(0), _ptr(twin), int(y_f16 is not None), _ptr(xhat), _ptr(rstd), eps,
            p_drop, seed, sync.data_ptr(), _gemm_ln_err_word(), _stream())
    rc = lib.icka_gemm_ln(*args)
    if rc == -1:        # ICKA_E_SHAPE: not a shape of the fused kernel
        return False
    check(rc, "icka_gemm_ln")
    if _PROF is not None:   # bench.py roofline leg: the launch is re-issued later, its operands must outlive the step
        d._keep = (h, w, o, bias, residual, gamma, beta, y, twin, xhat, rstd, sync)
        _PROF.append((2, args, 1, [d]))
    return True


def gemm_qkv_attn(x, w, bias, qkv, add_mask, ctx, lse, B, heads, S, *, p_drop=0.0, seed=0, scale=None, out16=None,
                  keepbits=None) -> bool:
    """QKV projection (x @ w^T + bias -> qkv, the stacked bf16 [q | k | v]; x / w bf16 or both fp16) and the whole-head
    self-attention of every (sample, head) as ONE launch (icka_gemm_qkv_attn) -- bitwise ``gemm(NT, x, w, qkv, bias=bias)``
    followed by ``attn_fwd`` on the three column blocks (``out16`` / ``keepbits`` as there).  Returns False (nothing launched)
    when the shape is not eligible: the caller then makes the two calls."""
    lib = _lib.load()
    d = gemm_desc(GEMM_NT, x, w, qkv, bias=bias)
    _mat(ctx, "ctx")
    H = heads * 64
    if tuple(qkv.shape) != (B * S, 3 * H) or tuple(ctx.shape) != (B * S, H):
        return False
    if add_mask.dtype != F32 or not add_mask.is_contiguous() or add_mask.numel() != B * S:
        raise ValueError("add_mask must be contiguous f32 [B, S]")
    if lse is not None and (lse.dtype != F32 or not lse.is_contiguous() or lse.numel() != B * heads * S):
        raise ValueError("lse must be contiguous f32 [B, heads, S]")
    if out16 is not None:
        _mat(out16, "out16", F16)
        if tuple(out16.shape) != tuple(ctx.shape) or out16.stride(0) != ctx.stride(0):
            raise ValueError("out16 must have the shape and row stride of ctx")
    if keepbits is not None and (not keepbits.is_cuda or keepbits.element_size() != 4 or not keepbits.is_contiguous()
                                 or keepbits.numel() < lib.icka_attn_keepbits_words(B, heads, S, S)):
        raise ValueError("keepbits: contiguous 32-bit device buffer of icka_attn_keepbits_words words (attn_keepbits)")
    args = (d, add_mask.data_ptr(), ctx.data_ptr(), _ptr(out16), ctx.stride(0), _ptr(lse), B, heads, S,
            (1.0 / 8.0) if scale is None else scale, p_drop, seed, _ptr(keepbits), _stream())
    rc = lib.icka_gemm_qkv_attn(*args)
    if rc == -1:        # ICKA_E_SHAPE
        return False
    check(rc, "icka_gemm_qkv_attn")
    if _PROF is not None:
        d._keep = (x, w, bias, qkv, add_mask, ctx, lse, out16, keepbits)
        _PROF.append((3, args, 1, [d]))
    return True


def gemm_ln_check_error(where: str = "") -> None:
    """Raise if a fused dense + LayerNorm launch ever reported a stripe wait that gave up.  The word is pinned host memory the
    kernel writes with a system-scope store: the check is a plain host read, no device synchronisation, so it runs at every host
    touch-point (the next fused launch, GraphedStep / GraphedModule replays).  The word is cleared when the error is raised."""
    if _GEMM_LN_ERR_NP is not None and _GEMM_LN_ERR_NP[0] != 0:
        _GEMM_LN_ERR_NP[0] = 0
        raise GemmLnHandoffError(
            "icka_amd fused dense + LayerNorm%s: a block gave up waiting for the other blocks of its 128-row stripe -- the grid "
            "was not co-resident (another kernel held CUs: a collective on another stream, a second model, a partitioned GPU).  "
            "The rows it finished are NaN.  Reserve CUs (icka_lstm_set_reserved_cus) or set ICKA_FUSE_DENSE_LN=0."
            % ((" (" + where + ")") if where else ""))


def ln_bwd_slabs(dy, xhat, rstd, gamma, partials, *, dy2=None, dres=None, dx=None, p_drop=0.0, seed=0) -> int:
    """LayerNorm backward without the parameter-gradient finalize; returns the number of slabs left in ``partials``
    (slot 0 -> dgamma, slot 1 -> dbeta) for a slab_reduction."""
    lib = _lib.load()
    _mat(dy, "dy"); _mat(xhat, "xhat")
    M, H = dy.shape
    check(lib.icka_ln_bwd_slabs(dy.data_ptr(), dy.stride(0), _ptr(dy2), _ld(dy2), xhat.data_ptr(), rstd.data_ptr(),
                                gamma.data_ptr(), _ptr(dres), _ld(dres), _ptr(dx), _ld(dx), partials.data_ptr(), M, H,
                                p_drop, seed, _stream()), "icka_ln_bwd_slabs")
    return lib.icka_ln_bwd_nslab(M)


def ln_bwd_workspace(H: int, device) -> torch.Tensor:
    return torch.empty(_lib.load().icka_ln_bwd_workspace_floats(H), dtype=F32, device=device)


def ln_bwd(dy, xhat, rstd, gamma, *, dy2=None, dres=None, dx=None, dgamma=None, dbeta=None, dbias=None,
           partials=None, p_drop=0.0, seed=0, accumulate=True):
    lib = _lib.load()
    _mat(dy, "dy"); _mat(xhat, "xhat")
    M, H = dy.shape
    check(lib.icka_ln_bwd(dy.data_ptr(), dy.stride(0), _ptr(dy2), _ld(dy2), xhat.data_ptr(), rstd.data_ptr(),
                          gamma.data_ptr(), _ptr(dres), _ld(dres), _ptr(dx), _ld(dx), _ptr(dgamma), _ptr(dbeta),
                          _ptr(dbias), partials.data_ptr(), M, H, p_drop, seed, int(accumulate), _stream()),
          "icka_ln_bwd")


# ------------------------------------------------------------------------------------------------- embeddings
def embed_fwd(ids, token_type, word, pos, typ, gamma, beta, y, *, y_f32=None, y_f16=None, xhat=None, rstd=None, eps=1e-12,
              p_drop=0.0, seed=0):
    lib = _lib.load()
    _dev(ids, "ids")
    B, S = ids.shape
    H = word.shape[1]
    if y_f32 is not None and y_f16 is not None:
        raise ValueError("one twin copy: y_f32 or y_f16")
    twin = y_f16 if y_f16 is not None else y_f32
    if twin is not None and (twin.dtype != (F16 if y_f16 is not None else F32) or not twin.is_contiguous()):
        raise ValueError("twin output must be contiguous f32 (y_f32) / fp16 (y_f16)")
    fn = lib.icka_embed_fwd_h if y_f16 is not None else lib.icka_embed_fwd
    check(fn(ids.data_ptr(), _ptr(token_type), word.data_ptr(), pos.data_ptr(), typ.data_ptr(),
             gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(twin), _ptr(xhat), _ptr(rstd),
             B, S, H, word.shape[0], typ.shape[0], eps, p_drop, seed, _stream()), "icka_embed_fwd")
    return y


def embed_bwd(dy, ids, token_type, xhat, rstd, gamma, dword, dpos, dtype_, dgamma, dbeta, partials, *,
              padding_idx=0, p_drop=0.0, seed=0, accumulate=True):
    lib = _lib.load()
    B, S = ids.shape
    H = dword.shape[1]
    check(lib.icka_embed_bwd(dy.data_ptr(), ids.data_ptr(), _ptr(token_type), xhat.data_ptr(), rstd.data_ptr(),
                             gamma.data_ptr(), dword.data_ptr(), dpos.data_ptr(), dtype_.data_ptr(),
                             dgamma.data_ptr(), dbeta.data_ptr(), partials.data_ptr(), B, S, H, dword.shape[0],
                             dtype_.shape[0], padding_idx, p_drop, seed, int(accumulate), _stream()),
          "icka_embed_bwd")


def embed_bwd_rows(dy, ids, token_type, xhat, rstd, gamma, dtok, dpos, dtype_, dgamma, dbeta, partials, *, vocab,
                   padding_idx=0, p_drop=0.0, seed=0, accumulate=True):
    """embed_bwd with the word-table gradient left as per-token rows ``dtok`` f32 [B*S, H] (row-sparse data-parallel exchange)."""
    lib = _lib.load()
    B, S = ids.shape
    H = dtok.shape[1]
    if dtok.dtype != F32 or tuple(dtok.shape) != (B * S, H) or not dtok.is_contiguous():
        raise ValueError("dtok must be contiguous f32 [B*S, H]")
    check(lib.icka_embed_bwd_rows(dy.data_ptr(), ids.data_ptr(), _ptr(token_type), xhat.data_ptr(), rstd.data_ptr(),
                                  gamma.data_ptr(), dtok.data_ptr(), dpos.data_ptr(), dtype_.data_ptr(), dgamma.data_ptr(),
                                  dbeta.data_ptr(), partials.data_ptr(), B, S, H, int(vocab), dtype_.shape[0], padding_idx, p_drop,
                                  seed, int(accumulate), _stream()), "icka_embed_bwd_rows")


def embed_scatter_rows(rows, ids, dword, *, padding_idx=0, scale=1.0):
    """dword[ids[t]] += scale * rows[t]  (rows f32 or bf16 [T, H] contiguous, ids int64 [T], dword f32 [vocab, H])."""
    _dev(rows, "rows"); _dev(ids, "ids"); _dev(dword, "dword")
    if rows.dtype not in (F32, BF16) or rows.dim() != 2 or not rows.is_contiguous():
        raise ValueError("rows must be contiguous f32 / bf16 [T, H]")
    if ids.dtype != torch.int64 or ids.numel() != rows.shape[0] or not ids.is_contiguous():
        raise ValueError("ids must be contiguous int64 with one entry per row")
    if dword.dtype != F32 or dword.dim() != 2 or dword.shape[1] != rows.shape[1] or not dword.is_contiguous():
        raise ValueError("dword must be contiguous f32 [vocab, H]")
    check(_lib.load().icka_embed_scatter_rows(rows.data_ptr(), int(rows.dtype == BF16), ids.data_ptr(), dword.data_ptr(),
                                              rows.shape[0], rows.shape[1], dword.shape[0], padding_idx, float(scale), _stream()),
          "icka_embed_scatter_rows")


def embed_prompt_fwd(ids, src, prompt, word, pos, typ, gamma, beta, y, *, y_f32=None, xhat=None, rstd=None,
                     pos_offset=0, eps=1e-5, p_drop=0.0, seed=0):
    """Prompt-spliced embeddings + LayerNorm + dropout (icka_hip.h: icka_embed_prompt_fwd).  ids int64 [B,S_in],
    src int32 [S], prompt bf16 [B,P,H] contiguous, y bf16 [B*S,H]."""
    lib = _lib.load()
    _dev(ids, "ids"); _dev(src, "src"); _dev(prompt, "prompt")
    if ids.dtype != torch.int64 or not ids.is_contiguous() or ids.dim() != 2:
        raise ValueError("ids must be contiguous int64 [B,S_in]")
    if src.dtype != torch.int32 or src.dim() != 1 or not src.is_contiguous():
        raise ValueError("src must be contiguous int32 [S]")
    B, S_in = ids.shape
    S, H = src.shape[0], word.shape[1]
    if prompt.dtype != BF16 or prompt.dim() != 3 or prompt.shape[0] != B or prompt.shape[2] != H or not prompt.is_contiguous():
        raise ValueError("prompt must be contiguous bf16 [B,P,H]")
    if pos.shape[0] < S + pos_offset:
        raise ValueError("position table has %d rows, the spliced sequence needs %d" % (pos.shape[0], S + pos_offset))
    if tuple(y.shape) != (B * S, H):
        raise ValueError("y must be [B*S,H]")
    check(lib.icka_embed_prompt_fwd(ids.data_ptr(), src.data_ptr(), prompt.data_ptr(), word.data_ptr(), pos.data_ptr(),
                                    typ.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ptr(y_f32),
                                    _ptr(xhat), _ptr(rstd), B, S_in, S, prompt.shape[1], H, word.shape[0], pos_offset,
                                    eps, p_drop, seed, _stream()), "icka_embed_prompt_fwd")
    return y


def embed_prompt_bwd(dy, ids, src, xhat, rstd, gamma, dword, dpos, dtype_, dgamma, dbeta, dprompt, partials, *,
                     pos_offset=0, padding_idx=1, p_drop=0.0, seed=0, accumulate=True):
    lib = _lib.load()
    B, S_in = ids.shape
    S, H = src.shape[0], dword.shape[1]
    if dprompt.dtype != BF16 or not dprompt.is_contiguous() or dprompt.shape[0] != B or dprompt.shape[2] != H:
        raise ValueError("dprompt must be contiguous bf16 [B,P,H]")
    if partials.numel() < S * lib.icka_ln_slab_slots() * H:
        raise ValueError("partials too small")
    check(lib.icka_embed_prompt_bwd(dy.data_ptr(), ids.data_ptr(), src.data_ptr(), xhat.data_ptr(), rstd.data_ptr(),
                                    gamma.data_ptr(), dword.data_ptr(), dpos.data_ptr(), dtype_.data_ptr(),
                                    dgamma.data_ptr(), dbeta.data_ptr(), dprompt.data_ptr(), partials.data_ptr(), B,
                                    S_in, S, dprompt.shape[1], H, dword.shape[0], pos_offset, padding_idx, p_drop, seed,
                                    int(accumulate), _stream()), "icka_embed_prompt_bwd")


def tanh_bwd(dy, y, dx):
    """dx = dy * (1 - y^2), contiguous bf16."""
    for n, t in (("dy", dy), ("y", y), ("dx", dx)):
        _dev(t, n)
        if t.dtype != BF16 or not t.is_contiguous():
            raise ValueError("%s must be contiguous bf16" % n)
    if dy.numel() != y.numel() or dx.numel() != y.numel():
        raise ValueError("tanh_bwd: size mismatch")
    check(_lib.load().icka_tanh_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), y.numel(), _stream()), "icka_tanh_bwd")
    return dx


def dgelu(dg, z, dz):
    """dz = dg * gelu'(z), contiguous bf16."""
    for n, t in (("dg", dg), ("z", z), ("dz", dz)):
        _dev(t, n)
        if t.dtype != BF16 or not t.is_contiguous() or t.numel() != z.numel():
            raise ValueError("%s must be contiguous bf16 with %d elements" % (n, z.numel()))
    check(_lib.load().icka_dgelu_bf16(dg.data_ptr(), z.data_ptr(), dz.data_ptr(), z.numel(), _stream()), "icka_dgelu_bf16")
    return dz


# ------------------------------------------------------------------------------------------------- attention
def attn_keepbits(B, heads, Sq, Skv, device) -> torch.Tensor:
    """Buffer for the keep bits of an attention-probability dropout site (attn_fwd fills it, attn_bwd reads it)."""
    return torch.empty(_lib.load().icka_attn_keepbits_words(B, heads, Sq, Skv), dtype=torch.int32, device=device)


def attn_fwd(q, k, v, add_mask, out, lse, B, heads, Sq, Skv, *, p_drop=0.0, seed=0, scale=None, fp8=False, out16=None,
             keepbits=None, tiled=False):
    """q/k/v/out: 2-D row-major bf16 views [B*S, >=heads*64] (may be column slices of a fused projection).
    fp8=True: QK^T and PV on the fp8 matrix cores (Sq, Skv <= 128 only).  out16: optional fp16 copy of the context with
    the strides of ``out`` (the "mixed16" operand of the out-proj GEMM).  tiled=True takes the tiled flash-style kernels
    also where the head would fit the whole-head kernels (per call: tests exercise both paths)."""
    lib = _lib.load()
    for n, t in (("q", q), ("k", k), ("v", v), ("out", out)):
        _mat(t, n)
    if scale is None:
        scale = 1.0 / math.sqrt(64.0)
    if out16 is not None or keepbits is not None or tiled:
        if out16 is not None:
            _mat(out16, "out16", F16)
            if tuple(out16.shape) != tuple(out.shape) or out16.stride(0) != out.stride(0):
                raise ValueError("out16 must have the shape and row stride of out")
        if keepbits is not None and (not keepbits.is_cuda or keepbits.element_size() != 4 or not keepbits.is_contiguous()
                                     or keepbits.numel() < lib.icka_attn_keepbits_words(B, heads, Sq, Skv)):
            raise ValueError("keepbits: contiguous 32-bit device buffer of icka_attn_keepbits_words words (attn_keepbits)")
        check(lib.icka_attn_fwd_ex(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                                   add_mask.data_ptr(), out.data_ptr(), _ptr(out16), out.stride(0), _ptr(lse), B, heads,
                                   Sq, Skv, scale, p_drop, seed, (_lib.ATTN_FP8 if fp8 else 0) | (_lib.ATTN_TILED if tiled else 0),
                                   _ptr(keepbits), _stream()), "icka_attn_fwd_ex")
        return out
    fn = lib.icka_attn_fwd_fp8 if fp8 else lib.icka_attn_fwd
    check(fn(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0), add_mask.data_ptr(),
             out.data_ptr(), out.stride(0), _ptr(lse), B, heads, Sq, Skv, scale, p_drop, seed, _stream()),
          "icka_attn_fwd_fp8" if fp8 else "icka_attn_fwd")
    return out


def attn_bwd(q, k, v, add_mask, out, dout, lse, delta, dq, dk, dv, B, heads, Sq, Skv, *, p_drop=0.0, seed=0,
             scale=None, keepbits=None, tiled=False):
    lib = _lib.load()
    for n, t in (("q", q), ("k", k), ("v", v), ("out", out), ("dout", dout), ("dq", dq), ("dk", dk), ("dv", dv)):
        _mat(t, n)
    if scale is None:
        scale = 1.0 / math.sqrt(64.0)
    check(lib.icka_attn_bwd(q.data_ptr(), q.stride(0), k.data_ptr(), k.stride(0), v.data_ptr(), v.stride(0),
                            add_mask.data_ptr(), out.data_ptr(), out.stride(0), dout.data_ptr(), dout.stride(0),
                            lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dq.stride(0), dk.data_ptr(),
                            dk.stride(0), dv.data_ptr(), dv.stride(0), B, heads, Sq, Skv, scale, p_drop, seed,
                            _ptr(keepbits), _lib.ATTN_TILED if tiled else 0, _stream()), "icka_attn_bwd")


# ------------------------------------------------------------------------------------------------- LSTM
def lstm_fwd(gates_x, w_hh, y, c_all, act, hprev, B, S, H, flags=0):
    """gates_x f32 [B*S, 8H]; w_hh bf16 [8H, H] (= [2][4H][H]); y bf16 [B*S, 2H]; c_all f32 [B*S, 2H];
    act bf16 [B*S, 8H]; hprev bf16 [B*S, 2H] or None."""
    _dev(gates_x, "gates_x")
    if gates_x.dtype != F32 or gates_x.stride(1) != 1 or tuple(gates_x.shape) != (B * S, 8 * H):
        raise ValueError("gates_x must be f32 [B*S, 8H]")
    for n, t, shp, dt in (("w_hh", w_hh, (8 * H, H), BF16), ("y", y, (B * S, 2 * H), BF16),
                          ("c_all", c_all, (B * S, 2 * H), F32), ("act", act, (B * S, 8 * H), BF16)):
        _dev(t, n)
        if t.dtype != dt or tuple(t.shape) != shp or not t.is_contiguous():
            raise ValueError("%s must be contiguous %s %s" % (n, dt, shp))
    check(_lib.load().icka_lstm_fwd(gates_x.data_ptr(), gates_x.stride(0), w_hh.data_ptr(), y.data_ptr(),
                                    c_all.data_ptr(), act.data_ptr(), _ptr(hprev), B, S, H, int(flags), _stream()), "icka_lstm_fwd")


def lstm_bwd(dy, w_hh_t, act, c_all, dgates, dc_carry, B, S, H, flags=0):
    for n, t, shp, dt in (("dy", dy, (B * S, 2 * H), BF16), ("w_hh_t", w_hh_t, (2 * H, 4 * H), BF16),
                          ("act", act, (B * S, 8 * H), BF16), ("c_all", c_all, (B * S, 2 * H), F32),
                          ("dgates", dgates, (B * S, 8 * H), BF16), ("dc_carry", dc_carry, (2 * B, H), F32)):
        _dev(t, n)
        if t.dtype != dt or tuple(t.shape) != shp or not t.is_contiguous():
            raise ValueError("%s must be contiguous %s %s" % (n, dt, shp))
    check(_lib.load().icka_lstm_bwd(dy.data_ptr(), w_hh_t.data_ptr(), act.data_ptr(), c_all.data_ptr(),
                                    dgates.data_ptr(), dgates.stride(0), dc_carry.data_ptr(), B, S, H, int(flags), _stream()),
          "icka_lstm_bwd")


class LstmHandoffError(RuntimeError):
    """A hand-off wait of a persistent LSTM launch gave up: the outputs of that call are NaN-poisoned."""


def lstm_check_error(where: str = "") -> None:
    """Raise if a persistent LSTM launch ever reported a failed hand-off (icka_hip.h: icka_lstm_barrier_error).  The error
    word is host-mapped memory: the check is a plain host read, no device synchronisation, so it runs at every host
    touch-point (BiLSTM.forward, GraphedStep / SegmentedStep replays).  The word is cleared when the error is raised."""
    lib = _lib.load()
    rc = lib.icka_lstm_barrier_error()
    if rc != 0:
        lib.icka_lstm_clear_error()
        raise LstmHandoffError(
            "icka_amd BiLSTM%s: a block of a persistent LSTM launch gave up waiting for another block's words (status %d): "
            "the grid was not co-resident (another kernel held its CUs, e.g. a collective on the communication stream, or "
            "the GPU is partitioned / shared).  The outputs and gradients of that call are NaN.  Reserve CUs "
            "(icka_lstm_set_reserved_cus) or take the per-step launches (BiLSTM.recurrence_flags = LSTM_PER_STEP)."
            % ((" (" + where + ")") if where else "", rc))


def copy_many(srcs, dsts) -> None:
    """dsts[i] <- srcs[i] (contiguous device tensors of equal byte size, at most 8) in one launch."""
    import ctypes as C
    n = len(srcs)
    if n != len(dsts) or n > 8:
        raise ValueError("copy_many: up to 8 (source, destination) pairs")
    sp = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dp = (C.c_void_p * n)(*[t.data_ptr() for t in dsts])
    for a, b in zip(srcs, dsts):
        if a.numel() * a.element_size() != b.numel() * b.element_size() or not (a.is_contiguous() and b.is_contiguous()):
            raise ValueError("copy_many: contiguous tensors of equal byte size")
    nb = (C.c_int64 * n)(*[t.numel() * t.element_size() for t in srcs])
    check(_lib.load().icka_copy_many(sp, dp, nb, n, _stream()), "icka_copy_many")


_LSTM_RESERVED = [0]


def lstm_set_reserved_cus(n: int) -> int:
    """Set the CUs kept free of persistent BiLSTM blocks (process-global); returns the previous value so that the caller can
    restore it (dp.GradReducer.close)."""
    check(_lib.load().icka_lstm_set_reserved_cus(int(n)), "icka_lstm_set_reserved_cus")
    prev, _LSTM_RESERVED[0] = _LSTM_RESERVED[0], int(n)
    return prev


def linear_small_m(x, W, bias, y, act: int = 0):
    """y bf16 [M<=64, N] = act(x . W^T + bias); x may be a strided row view (the pooler's first-token rows)."""
    _mat(x, "x"); _mat(W, "W"); _mat(y, "y")
    M, Kd = x.shape
    N = W.shape[0]
    if W.shape[1] != Kd or tuple(y.shape) != (M, N) or not W.is_contiguous():
        raise ValueError("linear_small_m: W [N,K] contiguous, y [M,N]")
    check(_lib.load().icka_linear_small_m(x.data_ptr(), x.stride(0), W.data_ptr(), _ptr(bias), y.data_ptr(), y.stride(0),
                                          M, N, Kd, act, _stream()), "icka_linear_small_m")
    return y


def transpose_bf16(src, dst, batch, R, Cn):
    _dev(src, "src"); _dev(dst, "dst")
    if src.dtype != BF16 or dst.dtype != BF16 or src.numel() != batch * R * Cn or dst.numel() != src.numel() \
            or not src.is_contiguous() or not dst.is_contiguous():
        raise ValueError("transpose_bf16: contiguous bf16 [batch, R, C] -> [batch, C, R]")
    check(_lib.load().icka_transpose_bf16(src.data_ptr(), dst.data_ptr(), batch, R, Cn, _stream()), "icka_transpose_bf16")
    return dst


# ------------------------------------------------------------------------------------------------- CRF
def _crf_check(emissions, start, end, trans):
    for n, t in (("emissions", emissions), ("start", start), ("end", end), ("trans", trans)):
        _dev(t, n)
        if t.dtype != F32 or not t.is_contiguous():
            raise ValueError("%s must be contiguous f32" % n)
    if emissions.dim() != 3:
        raise ValueError("emissions must be [B,S,C]")
    B, S, Cn = emissions.shape
    if start.numel() != Cn or end.numel() != Cn or tuple(trans.shape) != (Cn, Cn):
        raise ValueError("transition parameters do not match num_tags = %d" % Cn)
    return B, S, Cn


def _i64(t, name, shape):
    if t is None:
        return None
    _dev(t, name)
    if t.dtype != torch.int64 or tuple(t.shape) != shape or not t.is_contiguous():
        raise ValueError("%s must be contiguous int64 %s" % (name, shape))
    return t


def crf_llh(emissions, tags, mask, start, end, trans, llh):
    B, S, Cn = _crf_check(emissions, start, end, trans)
    _i64(tags, "tags", (B, S)); _i64(mask, "mask", (B, S))
    check(_lib.load().icka_crf_llh(emissions.data_ptr(), Cn, tags.data_ptr(), _ptr(mask), start.data_ptr(),
                                   end.data_ptr(), trans.data_ptr(), llh.data_ptr(), B, S, Cn, _stream()),
          "icka_crf_llh")
    return llh


def crf_grad(emissions, tags, mask, start, end, trans, gllh, d_emissions, d_start, d_end, d_trans):
    B, S, Cn = _crf_check(emissions, start, end, trans)
    _i64(tags, "tags", (B, S)); _i64(mask, "mask", (B, S))
    check(_lib.load().icka_crf_grad(emissions.data_ptr(), Cn, tags.data_ptr(), _ptr(mask), start.data_ptr(),
                                    end.data_ptr(), trans.data_ptr(), gllh.data_ptr(), d_emissions.data_ptr(), Cn,
                                    d_start.data_ptr(), d_end.data_ptr(), d_trans.data_ptr(), B, S, Cn, _stream()),
          "icka_crf_grad")


def crf_decode(emissions, mask, start, end, trans, best_tags, best_score=None):
    B, S, Cn = _crf_check(emissions, start, end, trans)
    _i64(mask, "mask", (B, S)); _i64(best_tags, "best_tags", (B, S))
    check(_lib.load().icka_crf_decode(emissions.data_ptr(), Cn, _ptr(mask), start.data_ptr(), end.data_ptr(),
                                      trans.data_ptr(), best_tags.data_ptr(), _ptr(best_score), B, S, Cn, _stream()),
          "icka_crf_decode")
    return best_tags


# ------------------------------------------------------------------------------------------------- helpers
def cast_f32_to_bf16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    _dev(src, "src"); _dev(dst, "dst")
    check(_lib.load().icka_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "icka_cast")
    return dst


def cast_f32_to_bf16_f16(src: torch.Tensor, dst: torch.Tensor, dsth: torch.Tensor) -> None:
    """both 16-bit shadows of an f32 range in one launch: dst bf16, dsth fp16"""
    _dev(src, "src"); _dev(dst, "dst"); _dev(dsth, "dsth")
    if src.dtype != F32 or dst.dtype != BF16 or dsth.dtype != F16 or dst.numel() != src.numel() or dsth.numel() != src.numel():
        raise TypeError("cast_f32_to_bf16_f16: f32 source, bf16 + fp16 destinations of the same length")
    check(_lib.load().icka_cast_f32_to_bf16_f16(src.data_ptr(), dst.data_ptr(), dsth.data_ptr(), src.numel(), _stream()),
          "icka_cast_f32_to_bf16_f16")


def cast_to_f16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """contiguous bf16 / f32 -> fp16 (saturating)"""
    _dev(src, "src"); _dev(dst, "dst")
    if src.dtype not in (BF16, F32) or dst.dtype != F16 or not src.is_contiguous() or not dst.is_contiguous() \
            or src.numel() != dst.numel():
        raise TypeError("cast_to_f16: contiguous bf16/f32 source, contiguous fp16 destination of the same size")
    check(_lib.load().icka_cast_to_f16(src.data_ptr(), int(src.dtype == F32), dst.data_ptr(), src.numel(), _stream()),
          "icka_cast_to_f16")
    return dst


def cast_bf16_to_f32(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    _dev(src, "src"); _dev(dst, "dst")
    check(_lib.load().icka_cast_bf16_to_f32(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), "icka_cast")
    return dst


def cast_pad_f32_to_bf16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """dst bf16 [M, ldd>=N]: dst[:, :N] = src f32 [M,N]; pad columns zeroed."""
    _dev(src, "src"); _dev(dst, "dst")
    if src.dtype != F32 or src.dim() != 2 or src.stride(1) != 1 or dst.dtype != BF16 or not dst.is_contiguous():
        raise TypeError("cast_pad: src f32 [M,N] row-major, dst contiguous bf16 [M,ldd]")
    M, N = src.shape
    check(_lib.load().icka_cast_pad_f32_to_bf16(src.data_ptr(), src.stride(0), dst.data_ptr(), dst.shape[1], M, N,
                                                _stream()), "icka_cast_pad_f32_to_bf16")
    return dst


def additive_mask(mask: torch.Tensor, T: int, out: torch.Tensor) -> torch.Tensor:
    """mask int64 [B, >=T] (row-major) -> out f32 [B,T] = (1 - mask[:, :T]) * -10000."""
    _dev(mask, "mask")
    if mask.dtype != torch.int64 or mask.stride(1) != 1:
        raise TypeError("mask must be int64 row-major")
    check(_lib.load().icka_additive_mask(mask.data_ptr(), mask.stride(0), out.data_ptr(), mask.shape[0], T,
                                         _stream()), "icka_additive_mask")
    return out


def dropout(x, y, *, y2=None, p_drop=0.0, seed=0):
    _mat(x, "x"); _mat(y, "y")
    M, H = x.shape
    check(_lib.load().icka_dropout(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), _ptr(y2), _ld(y2), M, H,
                                   p_drop, seed, _stream()), "icka_dropout")
    return y


def dropout_mask(n: int, p_drop: float, seed: int, device) -> torch.Tensor:
    out = torch.empty(n, dtype=F32, device=device)
    check(_lib.load().icka_dropout_mask(out.data_ptr(), n, p_drop, seed, _stream()), "icka_dropout_mask")
    return out


def attn_dropout_mask(rows: int, Skv: int, p_drop: float, seed: int, device) -> torch.Tensor:
    """The keep-multiplier [rows, Skv] the attention kernels apply to the probabilities (two decisions per hash: NOT
    dropout_mask of the flat index); rows = (batch * heads + head) * Sq + query."""
    out = torch.empty(rows, Skv, dtype=F32, device=device)
    check(_lib.load().icka_attn_dropout_mask(out.data_ptr(), rows, Skv, p_drop, seed, _stream()), "icka_attn_dropout_mask")
    return out


def regions_to_tokens(src: torch.Tensor, dst: torch.Tensor, B: int, R: int, Cc: int, layout: int) -> torch.Tensor:
    _dev(src, "src")
    if src.dtype != F32 or not src.is_contiguous():
        raise TypeError("region features must be contiguous f32")
    check(_lib.load().icka_regions_to_tokens(src.data_ptr(), dst.data_ptr(), B, R, Cc, layout, _stream()),
          "icka_regions_to_tokens")
    return dst


def regions_to_tokens_h(src: torch.Tensor, dst: torch.Tensor, dst16: torch.Tensor, B: int, R: int, Cc: int, layout: int) -> None:
    """regions_to_tokens with an fp16 twin of the tokens (the "mixed16" operand of the region projection)."""
    _dev(src, "src"); _dev(dst, "dst"); _dev(dst16, "dst16")
    if src.dtype != F32 or not src.is_contiguous() or dst.dtype != BF16 or dst16.dtype != F16:
        raise TypeError("regions_to_tokens_h: contiguous f32 features, bf16 + fp16 token buffers")
    check(_lib.load().icka_regions_to_tokens_h(src.data_ptr(), dst.data_ptr(), dst16.data_ptr(), B, R, Cc, layout, _stream()),
          "icka_regions_to_tokens_h")


def colsum_workspace(N: int, device) -> torch.Tensor:
    return torch.empty(_lib.load().icka_colsum_workspace_floats(N), dtype=F32, device=device)


def colsum(x, out, partials, accumulate=True):
    _mat(x, "x")
    M, N = x.shape
    check(_lib.load().icka_colsum(x.data_ptr(), x.stride(0), out.data_ptr(), partials.data_ptr(), M, N,
                                  int(accumulate), _stream()), "icka_colsum")
    return out


def gate_bwd(dout, g, cross, du, dcross, dcross_in=None):
    _mat(dout, "dout"); _mat(cross, "cross")
    M, H = dout.shape
    check(_lib.load().icka_gate_bwd(dout.data_ptr(), dout.stride(0), g.data_ptr(), cross.data_ptr(), cross.stride(0),
                                    _ptr(dcross_in), _ld(dcross_in), du.data_ptr(), dcross.data_ptr(),
                                    dcross.stride(0), M, H, _stream()), "icka_gate_bwd")


def add_bf16(a, b, out):
    check(_lib.load().icka_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "icka_add")
    return out


def token_ce(logits, labels, mask, loss_sum, count, dlogits):
    """logits f32 [M,C]; labels/mask int64 [M]; dlogits bf16 [M, ldd>=C] (unscaled, see scale_by_ratio)."""
    M, Cn = logits.shape
    check(_lib.load().icka_token_ce(logits.data_ptr(), logits.stride(0), labels.data_ptr(), mask.data_ptr(),
                                    loss_sum.data_ptr(), count.data_ptr(), dlogits.data_ptr(), dlogits.stride(0),
                                    M, Cn, _stream()), "icka_token_ce")


def token_ce_fused(logits, labels, mask, stats, dlogits):
    """Token CE without accumulator fill / atomics: stats f32[3] <- (loss sum, #valid, mean loss); dlogits bf16 or f32
    [M, ldd>=C] unscaled."""
    M, Cn = logits.shape
    lib = _lib.load()
    part = torch.empty(lib.icka_token_ce_workspace_floats(M), dtype=F32, device=logits.device)
    check(lib.icka_token_ce_fused(logits.data_ptr(), logits.stride(0), labels.data_ptr(), mask.data_ptr(),
                                  stats.data_ptr(), part.data_ptr(), dlogits.data_ptr(), dlogits.stride(0),
                                  int(dlogits.dtype == F32), M, Cn, _stream()), "icka_token_ce_fused")


def zero_(t: torch.Tensor) -> torch.Tensor:
    """Clear a contiguous f32 device tensor with the library's fill kernel."""
    _dev(t, "t")
    if t.dtype != F32 or not t.is_contiguous():
        raise TypeError("zero_: contiguous f32 tensor")
    check(_lib.load().icka_zero_f32(t.data_ptr(), t.numel(), _stream()), "icka_zero_f32")
    return t


def zero_rows_(t: torch.Tensor, row0: int) -> None:
    """Clear rows [row0, end) of a contiguous 16-bit [rows, cols] matrix (the padding rows of a row-padded operand) with the
    library's fill kernel."""
    _dev(t, "t")
    if t.dim() != 2 or not t.is_contiguous() or t.element_size() != 2:
        raise TypeError("zero_rows_: contiguous 16-bit [rows, cols] matrix")
    n16 = (t.shape[0] - row0) * t.shape[1]
    if n16 <= 0:
        return
    if (row0 * t.shape[1]) % 2 or n16 % 2:
        raise ValueError("zero_rows_: the cleared range must be a whole number of 32-bit words")
    check(_lib.load().icka_zero_f32(t.data_ptr() + 2 * row0 * t.shape[1], n16 // 2, _stream()), "icka_zero_f32")


def scale_by_ratio(x, y, num=None, den=None):
    """y = x * num[0] / max(den[0], 1) with device scalars (None = 1)."""
    check(_lib.load().icka_scale_by_ratio(x.data_ptr(), y.data_ptr(), _ptr(num), _ptr(den), x.numel(), _stream()),
          "icka_scale_by_ratio")
    return y


def scalar_ratio(out, num, den):
    check(_lib.load().icka_scalar_ratio(out.data_ptr(), num.data_ptr(), den.data_ptr(), _stream()),
          "icka_scalar_ratio")
    return out


def set_dropout_nonce(words: Optional[torch.Tensor]) -> None:
    """Register (or clear) the 2 x int32 device tensor whose value every dropout kernel folds into its seed."""
    if words is not None and (not words.is_cuda or words.numel() < 2 or words.element_size() != 4):
        raise TypeError("nonce must be a device tensor of two 32-bit words")
    global _NONCE_PTR
    check(_lib.load().icka_set_dropout_nonce(_ptr(words)), "icka_set_dropout_nonce")
    _NONCE_PTR = _ptr(words)


_NONCE_PTR = None


def clear_dropout_nonce_if(words: torch.Tensor) -> None:
    """Unregister ``words`` if (and only if) it is the currently registered nonce (its owner is going away)."""
    if _NONCE_PTR is not None and words is not None and _NONCE_PTR == words.data_ptr():
        set_dropout_nonce(None)


def bump_dropout_nonce(words: torch.Tensor) -> None:
    check(_lib.load().icka_bump_dropout_nonce(words.data_ptr(), _stream()), "icka_bump_dropout_nonce")


# ------------------------------------------------------------------------------------------------- data parallel
class DpFlagError(RuntimeError):
    """A bucket-ready wait on the communication stream gave up: that step's all-reduce ran on unfinished gradients (the
    bucket was poisoned with a NaN)."""


def dp_chunk_table(ranges, device) -> torch.Tensor:
    """Chunk table for dp_cast_chunks: [(lo, hi)] element ranges (multiples of 8) -> int64 [n, 2] device tensor of
    (first element, count <= icka_dp_chunk_elems())."""
    ch = _lib.load().icka_dp_chunk_elems()
    rows = []
    for lo, hi in ranges:
        if lo % 8 or hi % 8:
            raise ValueError("chunk ranges must start and end on multiples of 8 elements")
        while lo < hi:
            n = min(ch, hi - lo)
            rows.append((lo, n))
            lo += n
    return torch.tensor(rows, dtype=torch.int64, device=device).reshape(-1, 2)


def dp_cast_chunks(src_f32: torch.Tensor, dst_bf16: torch.Tensor, table: torch.Tensor) -> None:
    """dst[i] = bf16(src[i]) over the chunks of ``table`` (dp_chunk_table); src / dst are the whole flat buffers."""
    _dev(src_f32, "src"); _dev(dst_bf16, "dst")
    if src_f32.dtype != F32 or dst_bf16.dtype != BF16 or src_f32.numel() != dst_bf16.numel():
        raise TypeError("dp_cast_chunks: f32 source and bf16 destination of the same length")
    if table.numel() == 0:
        return
    check(_lib.load().icka_dp_cast_chunks(src_f32.data_ptr(), dst_bf16.data_ptr(), table.data_ptr(), table.shape[0],
                                          _stream()), "icka_dp_cast_chunks")


def dp_cast_back_scaled(src_bf16: torch.Tensor, dst_f32: torch.Tensor, scale: float) -> None:
    _dev(src_bf16, "src"); _dev(dst_f32, "dst")
    if src_bf16.dtype != BF16 or dst_f32.dtype != F32 or src_bf16.numel() != dst_f32.numel():
        raise TypeError("dp_cast_back_scaled: bf16 source and f32 destination of the same length")
    check(_lib.load().icka_dp_cast_back_scaled(src_bf16.data_ptr(), dst_f32.data_ptr(), src_bf16.numel(), float(scale),
                                               _stream()), "icka_dp_cast_back_scaled")


def dp_check_error(where: str = "") -> None:
    lib = _lib.load()
    if lib.icka_dp_error() != 0:
        lib.icka_dp_clear_error()
        raise DpFlagError("icka_amd data-parallel step%s: a bucket-ready wait on the communication stream gave up (the "
                          "compute graph did not reach the bucket's flag in time); the first gradients of that bucket were "
                          "set to NaN after the exchange (icka_dp_poison_if / icka_dp_poison_final), so a global-norm clip or "
                          "an optimizer step that consumed them produced NaN" % ((" (" + where + ")") if where else ""))


# ------------------------------------------------------------------------------------------------- per-sample gates
def sample_gate_fwd(a, c, gate, mode, out, B, S):
    _mat(a, "a"); _mat(out, "out")
    H = a.shape[1]
    check(_lib.load().icka_sample_gate_fwd(a.data_ptr(), a.stride(0), _ptr(c), _ld(c), gate.data_ptr(), mode,
                                           out.data_ptr(), out.stride(0), B, S, H, _stream()), "icka_sample_gate_fwd")
    return out


def sample_gate_fwd_h(a16, gate, mode, out, out16, B, S):
    """out (bf16) and out16 (fp16) = g(gate[b]) * a16 (fp16): the "mixed16" form of the gate without blend operand."""
    _mat(a16, "a16", F16); _mat(out, "out"); _mat(out16, "out16", F16)
    H = a16.shape[1]
    if out.stride(0) != out16.stride(0):
        raise ValueError("out and out16 share one row stride")
    check(_lib.load().icka_sample_gate_fwd_h(a16.data_ptr(), a16.stride(0), gate.data_ptr(), mode, out.data_ptr(),
                                             out16.data_ptr(), out.stride(0), B, S, H, _stream()), "icka_sample_gate_fwd_h")
    return out


def sample_gate_bwd(dout, a, c, gate, mode, da, dc, dgate, B, S):
    _mat(dout, "dout"); _mat(a, "a"); _mat(da, "da")
    H = a.shape[1]
    check(_lib.load().icka_sample_gate_bwd(dout.data_ptr(), dout.stride(0), a.data_ptr(), a.stride(0), _ptr(c), _ld(c),
                                           gate.data_ptr(), mode, da.data_ptr(), da.stride(0), _ptr(dc), _ld(dc),
                                           dgate.data_ptr(), B, S, H, _stream()), "icka_sample_gate_bwd")


def crs_fwd(seq, cross, W, bias, crs, B, S):
    _mat(seq, "seq"); _mat(cross, "cross")
    H = seq.shape[1]
    check(_lib.load().icka_crs_fwd(seq.data_ptr(), seq.stride(0), cross.data_ptr(), cross.stride(0), W.data_ptr(),
                                   _ptr(bias), crs.data_ptr(), B, S, H, _stream()), "icka_crs_fwd")
    return crs


def crs_bwd(dcrs, seq, cross, W, dseq, dcross, dW, dbias, B, S, accumulate):
    H = seq.shape[1]
    check(_lib.load().icka_crs_bwd(dcrs.data_ptr(), seq.data_ptr(), seq.stride(0), cross.data_ptr(), cross.stride(0),
                                   W.data_ptr(), dseq.data_ptr(), dcross.data_ptr(), dW.data_ptr(), _ptr(dbias), B, S, H,
                                   int(accumulate), _stream()), "icka_crs_bwd")
