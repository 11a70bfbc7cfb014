"""ParamArena: one flat HBM allocation for all parameters of a module tree.

MI355X-first memory layout (288 GB HBM3E per GPU): instead of ~200 small parameter tensors we keep
  * ``flat``    fp32 master copy  -- the nn.Parameters are *views* into it (state_dict keys/shapes unchanged),
  * ``shadow``  bf16 copy of the same layout -- what the MFMA GEMMs read; refreshed by ONE cast launch at the start
                of every outermost forward (``shadow_policy = "always"``, the default: optimizers that update through
                ``p.data`` -- the reference's BertAdam, my_bert/optimization.py:153 -- leave no trace in the version
                counters), or only when a change was seen (``"tracked"``, see ``sync``),
  * ``gflat``   fp32 gradients      -- ``p.grad`` are views into it; backward kernels write straight into it
                (GEMM beta = 0/1), so data-parallel all-reduce runs over a few large contiguous slices.
Parameters are laid out in registration (= execution) order, each slot aligned to 8 elements (16 B in bf16), so the
reference's separate query/key/value Linear layers are physically one [3H,H] matrix (fused QKV GEMM) while their
state_dict entries stay separate.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

ALIGN = 8


class Slot(object):
    __slots__ = ("name", "off", "numel", "shape", "param", "live", "is_table")

    def __init__(self, name, off, numel, shape, param, is_table=False):
        self.name, self.off, self.numel, self.shape, self.param = name, off, numel, tuple(shape), param
        self.live = False   # gflat slot holds a valid (accumulating) gradient for the current cycle
        self.is_table = is_table   # nn.Embedding table: gathered from the f32 master, never a GEMM operand -> no shadow


class ParamArena(object):
    def __init__(self, root: nn.Module):
        named = []
        seen = set()
        for name, p in _collect(root, ""):
            if id(p) in seen:
                continue
            seen.add(id(p))
            named.append((name, p))
        if not named:
            raise ValueError("module has no parameters")
        dev = named[0][1].device
        for name, p in named:
            if p.device != dev:
                raise ValueError("all parameters must be on one device (%s is on %s, expected %s)" % (name, p.device, dev))
            if p.dtype != torch.float32:
                raise TypeError("parameters are fp32 masters (bf16 shadows are derived); %s is %s" % (name, p.dtype))
        off = 0
        self.slots: Dict[int, Slot] = {}
        self.order: List[Slot] = []
        tables = {id(m.weight) for m in root.modules() if isinstance(m, nn.Embedding)}
        for name, p in named:
            s = Slot(name, off, p.numel(), p.shape, p, id(p) in tables)
            self.slots[id(p)] = s
            self.order.append(s)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.device = dev
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        self.shadow16 = None   # fp16 shadow of the same layout: forward operands of the "mixed16" mode (enable_fp16_shadow)
        with torch.no_grad():
            for s in self.order:
                view = self.flat[s.off:s.off + s.numel].view(s.shape)
                view.copy_(s.param.data)
                s.param.data = view
        # element ranges of ``flat`` that get a bf16 shadow: everything but the embedding tables (20 % of the bert-base
        # path's parameters: the per-forward re-cast skips them); neighbouring ranges are merged across small gaps
        self._cast_ranges: List[Tuple[int, int]] = []
        for s in self.order:
            if s.is_table:
                continue
            lo, hi = s.off, s.off + (s.numel + ALIGN - 1) // ALIGN * ALIGN
            if self._cast_ranges and lo - self._cast_ranges[-1][1] < (1 << 16):
                self._cast_ranges[-1] = (self._cast_ranges[-1][0], hi)
            else:
                self._cast_ranges.append((lo, hi))
        self._synced = None
        self._synced_call = -1   # id of the outermost icka forward (ArenaModule.__call__) whose shadow refresh has run
        # "always": re-cast the bf16 shadow at every outermost forward (safe with ANY way of updating parameters);
        # "tracked": re-cast only when a parameter version counter moved, an optimizer step ran (global post-step
        # hook) or mark_dirty() was called -- the caller promises not to write parameters through ``.data``
        self.shadow_policy = "always"
        self._ws: Dict[Tuple[str, int], torch.Tensor] = {}
        # views of the flat buffers per parameter (group): the storage never moves while the arena lives, and building a
        # slice + view pair costs ~2-3 us of host time -- ~10 of them per GEMM call in an eager step (tools/eager_profile.py)
        self._vc: Dict[tuple, torch.Tensor] = {}
        # autograd anchor: a leaf that requires grad, passed to every Function so that backward runs even when
        # no *tensor input* requires grad (parameters are read from the arena, not passed through autograd)
        self.anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        self.keep_saved = False      # graph.GraphedModule: the backward of one recorded forward is captured twice (overwrite and
                                     # accumulate forms): the layer Functions keep their saved state across backward runs
        self.reducer = None          # optional dp.GradReducer: overlaps bucket all-reduces with backward
        self._pending_final: List[Slot] = []
        self.pending_wgrad = []      # queued weight-gradient GEMM descriptors (+ keep-alive tensors), see ops._wgrad
        self.pending_reductions = []  # LayerNorm dgamma/dbeta slab reductions riding on the next grouped launch
        self._seed_base = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        self._seed_ctr = 0
        self.seed_log = None     # tests: set to a list to record every seed handed out (one per dropout site call, in order)

    # ------------------------------------------------------------------------------------------ dropout seeds
    def set_seed(self, seed: int) -> None:
        self._seed_base, self._seed_ctr = seed & 0xFFFFFFFFFFFFFFFF, 0

    def next_seed(self) -> int:
        """splitmix64 stream: a fresh 64-bit seed per dropout site call; forward stores it for its backward."""
        self._seed_ctr += 1
        z = (self._seed_base + self._seed_ctr * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        if self.seed_log is not None:
            self.seed_log.append(z)
        return z

    # ------------------------------------------------------------------------------------------ validity
    def valid_for(self, root: nn.Module) -> bool:
        """False when a parameter was re-allocated behind our back (.to(), .cuda(), new Parameter objects).
        Cheap by design (runs in every module forward): .to()/.cuda() move ALL parameters, so checking the first
        one of the sub-tree is enough."""
        p = _first_param(root)      # (root.parameters() builds the whole named_modules machinery: ~3 us per module forward)
        if p is None:
            return False
        s = self.slots.get(id(p))
        return s is not None and p.data_ptr() == self.flat.data_ptr() + 4 * s.off

    # ------------------------------------------------------------------------------------------ views
    def slot(self, p: nn.Parameter) -> Slot:
        return self.slots[id(p)]

    def w(self, p: nn.Parameter) -> torch.Tensor:
        """bf16 shadow view of a parameter."""
        key = ("w", id(p))
        v = self._vc.get(key)
        if v is None:
            s = self.slots[id(p)]
            if s.is_table:
                raise RuntimeError("%s is an embedding table: it has no bf16 shadow (kernels gather the f32 master)" % s.name)
            v = self._vc[key] = self.shadow[s.off:s.off + s.numel].view(s.shape)
        return v

    # -- "mixed16": fp16 shadow for the forward GEMMs (the bf16 shadow keeps serving dgrad in backward)
    def enable_fp16_shadow(self) -> None:
        if self.shadow16 is None:
            if self.shadow is None:
                raise RuntimeError("ParamArena.enable_fp16_shadow: needs a ROCm device (no CPU path)")
            self.shadow16 = torch.zeros(self.total, dtype=torch.float16, device=self.device)
            self._synced = None   # the next sync fills it

    def w16(self, p: nn.Parameter) -> torch.Tensor:
        """fp16 shadow view of a parameter (enable_fp16_shadow first)."""
        key = ("w16", id(p))
        v = self._vc.get(key)
        if v is None:
            s = self.slots[id(p)]
            if s.is_table or self.shadow16 is None:
                raise RuntimeError("%s has no fp16 shadow (embedding table, or enable_fp16_shadow() was not called)" % s.name)
            v = self._vc[key] = self.shadow16[s.off:s.off + s.numel].view(s.shape)
        return v

    def w16_cat(self, ps: Sequence[nn.Parameter]) -> torch.Tensor:
        key = ("w16c",) + tuple(id(p) for p in ps)
        v = self._vc.get(key)
        if v is None:
            first, rows = self._adjacent(ps)
            if self.shadow16 is None:
                raise RuntimeError("enable_fp16_shadow() was not called")
            v = self._vc[key] = self.shadow16[first.off:first.off + rows * first.shape[-1]].view(rows, first.shape[-1])
        return v

    def w_cat(self, ps: Sequence[nn.Parameter]) -> torch.Tensor:
        """bf16 view of several 2-D [o_i, in] parameters as ONE [sum o_i, in] matrix (must be adjacent slots)."""
        key = ("wc",) + tuple(id(p) for p in ps)
        v = self._vc.get(key)
        if v is None:
            first, rows = self._adjacent(ps)
            v = self._vc[key] = self.shadow[first.off:first.off + rows * first.shape[-1]].view(rows, first.shape[-1])
        return v

    def f_cat(self, ps: Sequence[nn.Parameter]) -> torch.Tensor:
        """fp32 master view of adjacent 1-D parameters as one vector (fused biases)."""
        key = ("fc",) + tuple(id(p) for p in ps)
        v = self._vc.get(key)
        if v is None:
            first, n = self._adjacent(ps)
            v = self._vc[key] = self.flat[first.off:first.off + n]
        return v

    def g(self, p: nn.Parameter) -> torch.Tensor:
        key = ("g", id(p))
        v = self._vc.get(key)
        if v is None:
            s = self.slots[id(p)]
            v = self._vc[key] = self.gflat[s.off:s.off + s.numel].view(s.shape)
        return v

    def g_cat(self, ps: Sequence[nn.Parameter]) -> torch.Tensor:
        key = ("gc",) + tuple(id(p) for p in ps)
        v = self._vc.get(key)
        if v is None:
            first, rows = self._adjacent(ps)
            if len(first.shape) == 2:
                v = self.gflat[first.off:first.off + rows * first.shape[-1]].view(rows, first.shape[-1])
            else:
                v = self.gflat[first.off:first.off + rows]
            self._vc[key] = v
        return v

    def wire_of(self, gview: torch.Tensor, beta: float = 0.0):
        """Data parallel with bf16 buckets (dp.GradReducer): ``{"out3": wire, "out3_only": only}`` for kernels.gemm_desc --
        the slice of the reducer's bf16 wire buffer that mirrors the gradient view ``gview`` (a contiguous view of ``gflat``,
        e.g. ``g(p)`` / ``g_cat(ps)``), handed to a weight-gradient GEMM as its second output (icka_gemm_desc.C3) so that the
        wire copy comes out of the GEMM epilogue, and whether the GEMM may skip the f32 store altogether (GradReducer.wire_view;
        never when it accumulates, ``beta`` != 0).  Empty when there is no reducer, it exchanges f32, or the view is not
        contiguous."""
        r = self.reducer
        if r is None or getattr(r, "gwire", None) is None or not gview.is_contiguous():
            return {}
        off = (gview.data_ptr() - self.gflat.data_ptr()) // 4
        if off < 0 or off + gview.numel() > self.total:
            return {}
        wire, only = r.wire_view(off, gview.numel(), gview.shape)
        return {"out3": wire, "out3_only": bool(only and beta == 0.0)}

    def _adjacent(self, ps: Sequence[nn.Parameter]):
        sl = [self.slots[id(p)] for p in ps]
        rows = 0
        for a, b in zip(sl[:-1], sl[1:]):
            if a.off + a.numel != b.off or a.shape[1:] != b.shape[1:]:
                raise RuntimeError("parameters %s / %s are not adjacent in the arena" % (a.name, b.name))
        for s in sl:
            rows += s.shape[0]
        return sl[0], rows

    # ------------------------------------------------------------------------------------------ bf16 shadow
    def sync(self, force: bool = False) -> None:
        """Refresh the bf16 shadow (one cast launch).  Policy "always": every call casts.  Policy "tracked": only
        when a version counter moved / an optimizer stepped / mark_dirty() was called since the last refresh."""
        if self.shadow is None:
            raise RuntimeError("ParamArena.sync: bf16 shadows need a ROCm device (no CPU path)")
        v = _OPT_STEPS[0]
        for s in self.order:
            v += s.param._version
        if force or self.shadow_policy == "always" or v != self._synced:
            from . import kernels
            for lo, hi in self._cast_ranges:
                if self.shadow16 is not None:
                    kernels.cast_f32_to_bf16_f16(self.flat[lo:hi], self.shadow[lo:hi], self.shadow16[lo:hi])
                else:
                    kernels.cast_f32_to_bf16(self.flat[lo:hi], self.shadow[lo:hi])
            self._synced = v

    def mark_dirty(self) -> None:
        self._synced = None

    # ------------------------------------------------------------------------------------------ gradients
    def begin_step(self) -> None:
        """Kept for callers of the round-1 API; the overwrite-vs-accumulate decision is taken per slot at backward
        time (grad_beta), so nothing has to happen at the start of a forward."""

    def _is_live(self, s: Slot) -> bool:
        """A slot accumulates only if it was written in this accumulation cycle AND the user still holds that
        gradient: ``zero_grad()`` (p.grad = None) or a foreign ``p.grad`` tensor at ANY point before this backward
        -- including between forward and backward -- starts a fresh cycle."""
        g = s.param.grad
        return s.live and g is not None and g.data_ptr() == self.gflat.data_ptr() + 4 * s.off

    def grad_beta(self, ps) -> float:
        """beta for a gradient write into the slot(s): 0.0 on the first write of an accumulation cycle, else 1.0.
        Also (re)attaches p.grad to the arena view.  In a fused group with mixed state (one of q/k/v frozen or
        cleared by hand) the fresh slots are zeroed and the group accumulates."""
        if isinstance(ps, nn.Parameter):
            ps = (ps,)
        sl = [self.slots[id(p)] for p in ps]
        live = [self._is_live(s) for s in sl]
        mixed = any(live) and not all(live)
        for s, lv in zip(sl, live):
            if not lv:
                if mixed:
                    self.gflat[s.off:s.off + s.numel].zero_()
                s.live = True
                s.param.grad = self.g(s.param)
            self._pending_final.append(s)
        return 1.0 if live[0] or mixed else 0.0

    def flush_final(self) -> None:
        """End of a block's backward: every gradient slot written since the last flush is final for this step."""
        if self.reducer is not None and self._pending_final:
            self.reducer.mark_final(self._pending_final)
        self._pending_final = []

    def zero_grad(self) -> None:
        self.gflat.zero_()
        for s in self.order:
            s.live = False

    def attach_grads(self, slots=None) -> None:
        """(Re)attach ``p.grad`` views for slots whose gradient was produced without Python running (hipGraph replay)."""
        for s in (self.order if slots is None else slots):
            g = s.param.grad
            if g is None or g.data_ptr() != self.gflat.data_ptr() + 4 * s.off:
                s.param.grad = self.g(s.param)
            s.live = True

    # ------------------------------------------------------------------------------------------ workspace
    def workspace(self, tag: str, numel: int, dtype=torch.float32) -> torch.Tensor:
        key = (tag, numel)
        t = self._ws.get(key)
        if t is None or t.dtype != dtype:
            t = torch.empty(numel, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    # ------------------------------------------------------------------------------------------ DP buckets
    def buckets(self, bucket_elems: int) -> List[Tuple[int, int]]:
        """Contiguous [start, end) element ranges of gflat, from the END of the arena towards the start (the order
        in which backward produces gradients), each at least ``bucket_elems`` long (last one takes the rest).  A run
        of embedding tables starts a bucket of its own: their gradient is final only at the very end of backward and
        must not hold back the reduction of the encoder layers that share its tail of the arena."""
        out = []
        end = self.total
        cur = end
        prev_table = None
        for s in reversed(self.order):
            if prev_table is False and s.is_table and end > cur:
                out.append((cur, end))
                end = cur
            prev_table = s.is_table
            cur = s.off
            if end - cur >= bucket_elems:
                out.append((cur, end))
                end = cur
        if end > 0:
            out.append((0, end))
        return out


# number of optimizer steps seen by the global post-step hook: part of the "tracked" shadow fingerprint
_OPT_STEPS = [0]


def _install_optimizer_hook() -> None:
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:   # pragma: no cover
        return

    def _bump(optimizer, args, kwargs):
        _OPT_STEPS[0] += 1

    register_optimizer_step_post_hook(_bump)


_install_optimizer_hook()


def _first_param(module: nn.Module):
    """First parameter of a module tree in ``parameters()`` order (own parameters, then children), without generators."""
    for p in module._parameters.values():
        if p is not None:
            return p
    for child in module._modules.values():
        if child is not None:
            p = _first_param(child)
            if p is not None:
                return p
    return None


def _collect(module: nn.Module, prefix: str):
    """Parameters in arena order: registration order, except that a module may impose the order of its whole
    sub-tree through ``icka_param_order()`` (attention blocks put query/key/value weights, then their biases,
    back to back so that they form one fused [3H,H] operand)."""
    order = getattr(module, "icka_param_order", None)
    if order is not None:
        for name, p in order():
            yield prefix + name, p
        return
    for name, p in module._parameters.items():
        if p is not None:
            yield prefix + name, p
    for cname, child in module._modules.items():
        if child is not None:
            yield from _collect(child, prefix + cname + ".")


_FWD_DEPTH = [0]   # nesting depth of ArenaModule forwards (one Python thread per process)
_CALL_ID = [0]     # bumped whenever an OUTERMOST ArenaModule forward starts


def refresh_shadow_once(A: "ParamArena") -> None:
    """The shadow refresh of the current outermost icka forward: ``A.sync()`` exactly once per outermost call and arena,
    at whatever depth the first kernel-owning module of the call tree sits (a container such as BertAttention, whose own
    forward launches nothing, leaves it to its children; a trunk function and the BertModel it calls share one).  Outside
    any ArenaModule forward (free functions such as scalar_gate_fusion called on their own) every call refreshes."""
    if _FWD_DEPTH[0] == 0 or A._synced_call != _CALL_ID[0] or A._synced is None:
        A.sync()
        A._synced_call = _CALL_ID[0]


class ArenaModule(nn.Module):
    """Base of every icka module that owns parameters read by kernels.  ``_arena()`` returns the (shared) ParamArena
    and, ONCE per outermost icka forward of a call tree, refreshes the bf16 shadow (ParamArena.sync; with the "tracked"
    policy the version / optimizer-step fingerprint decides whether anything is cast)."""

    def __call__(self, *args, **kwargs):
        if _FWD_DEPTH[0] == 0:
            _CALL_ID[0] += 1
            # the OUTERMOST module of a call tree owns the arena of its whole sub-tree, also when its own forward launches
            # nothing (BertAttention, BertCrossAttention: containers whose children would otherwise build one arena each)
            if getattr(self, "_icka_arena", None) is None:
                p = _first_param(self)
                if p is not None and p.device.type == "cuda":
                    arena_of(self)
        _FWD_DEPTH[0] += 1
        try:
            return super().__call__(*args, **kwargs)
        finally:
            _FWD_DEPTH[0] -= 1

    def _arena(self) -> ParamArena:
        A = arena_of(self)
        if A.device.type != "cuda":
            raise RuntimeError("%s: parameters are on %s; move the module to a ROCm device (icka_amd has no CPU "
                               "path)" % (type(self).__name__, A.device))
        refresh_shadow_once(A)
        return A


def arena_of(module: nn.Module) -> ParamArena:
    """Get (or lazily build / rebuild) the arena that owns ``module``'s parameters.  The arena is created by the
    outermost icka module whose forward runs first and is shared with all its sub-modules."""
    a = getattr(module, "_icka_arena", None)
    if a is not None and a.valid_for(module):
        return a
    a = ParamArena(module)
    for m in module.modules():
        object.__setattr__(m, "_icka_arena", a)
    return a
