"""ArenaAdamW: the parameter update of the reference's training loop on the flat ParamArena buffers.

The reference updates with ``torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0); optimizer.step(); scheduler.step();
model.zero_grad()`` (My_cross_attention.py:831-844), ``optimizer = AdamW(grouped parameters, lr, weight_decay=0.01)`` with
biases and LayerNorm parameters in a no-decay group (:743-751).  As ~220 per-tensor launches that update costs 2.9 ms on top
of the 4.5 ms forward + backward of the c2 step.  Parameters, gradients (and here the two moment buffers) share ONE flat
layout, so the same update is three launches over chunk tables (include/icka_hip.h: icka_optim_*): gradient norm, clip
coefficient (kept on the device: no host synchronisation), and the AdamW arithmetic -- which also writes the 16-bit weight
shadows the next forward's GEMMs read, so the arena's per-forward re-cast has nothing left to do.

A ``torch.optim.Optimizer``: ``param_groups`` / ``lr`` are the usual ones, so the reference's
``get_linear_schedule_with_warmup`` (a LambdaLR) drives it unchanged.  Arithmetic = ``torch.optim.AdamW`` (what transformers
now ships in place of the ``transformers.AdamW`` the reference imported; that class is absent from the installed
transformers 5.x -- PARITY WITH IT IS UNPINNED: the defaults below (eps 1e-8, decoupled decay applied as in torch) are
torch's, the reference passed no eps; tests pin this class against torch.optim.AdamW + clip_grad_norm_).

Checkpoints: the moments and the step count live in flat arena-layout buffers, not in ``Optimizer.state``;
``state_dict()`` / ``load_state_dict()`` carry them (keys ``icka_m`` / ``icka_v`` / ``icka_t`` / ``icka_layout``), and a
resumed run continues with the same bias correction.  A state saved for another arena layout is refused; a model whose arena
is rebuilt after the first step (``.to()``, new Parameter objects) raises instead of silently restarting the moments.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import kernels as K
from .arena import _OPT_STEPS, ParamArena, arena_of

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")     # My_cross_attention.py:744


def reference_param_groups(model: torch.nn.Module, weight_decay: float = 0.01) -> List[dict]:
    """The reference's two groups (My_cross_attention.py:743-748): everything whose name contains 'bias', 'LayerNorm.bias'
    or 'LayerNorm.weight' is not decayed."""
    named = list(model.named_parameters())
    return [{"params": [p for n, p in named if not any(nd in n for nd in NO_DECAY)], "weight_decay": weight_decay},
            {"params": [p for n, p in named if any(nd in n for nd in NO_DECAY)], "weight_decay": 0.0}]


class ArenaAdamW(torch.optim.Optimizer):
    def __init__(self, model: torch.nn.Module, params=None, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, max_grad_norm: Optional[float] = None):
        """``model``: the module whose forward built (or will build) the ParamArena.  ``params``: parameters or param groups
        (default: ``reference_param_groups(model, weight_decay)``).  ``max_grad_norm``: clip the global gradient norm inside
        ``step()`` (the reference's clip_grad_norm_(..., 1.0)); None = no clipping."""
        if params is None:
            params = reference_param_groups(model, weight_decay)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.max_grad_norm = max_grad_norm
        self._arena: Optional[ParamArena] = None
        self._m = self._v = None
        self._t = 0
        self._sig = None
        self._tables = None
        self._norm_table = None
        self._partials = None
        self._clip = None
        self._pending = None     # state loaded before the arena was bound

    # ------------------------------------------------------------------------------------------------------------
    def _bind(self) -> ParamArena:
        A = arena_of(self.model)
        if A.device.type != "cuda":
            raise RuntimeError("ArenaAdamW: the model's parameters are on %s (icka_amd has no CPU path)" % A.device)
        if A is not self._arena:
            if self._arena is not None and self._t > 0:
                raise RuntimeError("ArenaAdamW: the model's parameter arena was rebuilt after %d optimizer steps (.to() / new "
                                   "Parameter objects): the moment buffers belong to the old layout.  Save state_dict() before "
                                   "moving the model and load it into a new optimizer afterwards." % self._t)
            self._arena = A
            self._m = torch.zeros(A.total, dtype=torch.float32, device=A.device)
            self._v = torch.zeros(A.total, dtype=torch.float32, device=A.device)
            self._clip = torch.ones(2, dtype=torch.float32, device=A.device)
            self._sig = None
            if self._pending is not None:       # a state loaded before the arena existed
                self._install(A, self._pending)
                self._pending = None
        return A

    # ------------------------------------------------------------------------------------------------------------ checkpoints
    @staticmethod
    def _layout(A: ParamArena):
        return [(s.name, s.off, s.numel) for s in A.order]

    def state_dict(self):
        sd = super().state_dict()
        if self._arena is not None and self._m is not None:
            sd["icka_m"], sd["icka_v"] = self._m.detach().cpu().clone(), self._v.detach().cpu().clone()
            sd["icka_layout"] = self._layout(self._arena)
        elif self._pending is not None:
            sd["icka_m"], sd["icka_v"], sd["icka_layout"] = self._pending["icka_m"], self._pending["icka_v"], self._pending["icka_layout"]
        sd["icka_t"] = int(self._t)
        return sd

    def load_state_dict(self, state_dict):
        extra = {k: state_dict[k] for k in ("icka_m", "icka_v", "icka_layout", "icka_t") if k in state_dict}
        if "icka_t" not in extra:
            raise ValueError("ArenaAdamW.load_state_dict: no 'icka_t' -- not a state_dict of this class (moments and step count "
                             "would silently restart)")
        super().load_state_dict({k: v for k, v in state_dict.items() if not k.startswith("icka_")})
        self._t = int(extra["icka_t"])
        if "icka_m" in extra:
            if self._arena is not None:
                self._install(self._arena, extra)
            else:
                self._pending = extra          # installed when the arena is bound (first step)

    def _install(self, A: ParamArena, extra) -> None:
        if list(map(tuple, extra["icka_layout"])) != self._layout(A):
            raise ValueError("ArenaAdamW.load_state_dict: the saved moments belong to another parameter layout (different model "
                             "or registration order)")
        self._m.copy_(extra["icka_m"])
        self._v.copy_(extra["icka_v"])

    def _build(self, A: ParamArena):
        """Chunk tables: one per param group over the slots that hold a gradient this step, one over all of them."""
        sig, per_group, every = [], [], []
        for gi, group in enumerate(self.param_groups):
            ranges = []
            for p in group["params"]:
                s = A.slots.get(id(p))
                if s is None:
                    raise RuntimeError("ArenaAdamW: a parameter of group %d is not in the model's arena" % gi)
                g = p.grad
                if g is None:
                    continue
                if g.data_ptr() != A.gflat.data_ptr() + 4 * s.off:     # a foreign gradient tensor: bring it into the arena
                    A.gflat[s.off:s.off + s.numel].view(s.shape).copy_(g)
                lo, hi = s.off, s.off + (s.numel + 7) // 8 * 8
                if ranges and ranges[-1][1] == lo:
                    ranges[-1] = (ranges[-1][0], hi)
                else:
                    ranges.append((lo, hi))
                sig.append(s.off)
            per_group.append(ranges)
            every += ranges
        sig = tuple(sig)
        if sig != self._sig:
            self._sig = sig
            self._tables = [K.dp_chunk_table(sorted(r), A.device) for r in per_group]
            self._norm_table = K.dp_chunk_table(sorted(every), A.device)
            self._partials = torch.empty(max(1, self._norm_table.shape[0]), dtype=torch.float32, device=A.device)

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        A = self._bind()
        self._build(A)
        if self._norm_table.shape[0] == 0:
            return loss
        lib = K._lib.load()
        st = K._stream()
        self._t += 1
        coef = None
        if self.max_grad_norm is not None:
            K.check(lib.icka_optim_sqnorm(A.gflat.data_ptr(), self._norm_table.data_ptr(), self._norm_table.shape[0],
                                          self._partials.data_ptr(), st), "icka_optim_sqnorm")
            K.check(lib.icka_optim_clip(self._partials.data_ptr(), self._norm_table.shape[0], float(self.max_grad_norm),
                                        self._clip.data_ptr(), st), "icka_optim_clip")
            coef = self._clip[1:].data_ptr()
        sh = A.shadow.data_ptr() if A.shadow is not None else None
        sh16 = A.shadow16.data_ptr() if A.shadow16 is not None else None
        for group, table in zip(self.param_groups, self._tables):
            if table.shape[0] == 0:
                continue
            b1, b2 = group["betas"]
            K.check(lib.icka_optim_adamw(A.flat.data_ptr(), A.gflat.data_ptr(), self._m.data_ptr(), self._v.data_ptr(), sh, sh16,
                                         table.data_ptr(), table.shape[0], coef, float(group["lr"]), float(b1), float(b2),
                                         float(group["eps"]), float(group["weight_decay"]), self._t, st), "icka_optim_adamw")
        # every GEMM operand that changed got its 16-bit shadows from the update kernel itself (embedding tables have none):
        # tell the arena, so that the "tracked" policy does not re-cast after this step (the global post-step hook bumps
        # _OPT_STEPS by one right after this method returns; parameters not in any group / without gradient did not change)
        v = _OPT_STEPS[0] + 1
        for s in A.order:
            v += s.param._version
        A._synced = v
        return loss

    def grad_norm(self) -> torch.Tensor:
        """Total gradient norm of the last clipped step (device scalar, as clip_grad_norm_ returns it)."""
        return self._clip[0]
