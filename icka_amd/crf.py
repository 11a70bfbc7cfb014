"""Linear-chain CRF layer on the HIP kernels (`icka_crf_*`).

Drop-in for the ``torchcrf.CRF`` object the reference constructs as ``CRF(num_tags, batch_first=True)``
(Cross_Modal_Interaction_Module.py:911; my_bert/cl_modeling.py:1269) and calls as
``-crf(emissions, tags=labels, mask=mask, reduction='token_mean' | 'mean')`` and ``crf.decode(emissions, mask=mask)``
(:1045-1057; cl_modeling.py:1380-1386): same constructor, parameter names (``start_transitions``, ``end_transitions``,
``transitions``), initialisation (uniform(-0.1, 0.1)), argument validation and return types.  The arithmetic follows
pytorch-crf 0.7.2 (the package is third-party and absent from the reference tree: parity unpinned, see
oracle/crf_oracle.py)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import kernels as K
from .arena import ArenaModule, ParamArena, arena_of

F32 = torch.float32


class _CrfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, emissions, tags, mask, mod, A: ParamArena):
        B = emissions.shape[0]
        llh = torch.empty(B, dtype=F32, device=emissions.device)
        K.crf_llh(emissions, tags, mask, mod.start_transitions, mod.end_transitions, mod.transitions, llh)
        ctx.mod, ctx.A = mod, A
        ctx.save_for_backward(emissions, tags, mask)
        return llh

    @staticmethod
    def backward(ctx, gllh):
        emissions, tags, mask = ctx.saved_tensors
        mod, A = ctx.mod, ctx.A
        ps = (mod.start_transitions, mod.end_transitions, mod.transitions)
        if A.grad_beta(ps) == 0.0:          # first write of this accumulation cycle: the kernel adds with atomics
            for p in ps:
                A.g(p).zero_()
        de = torch.empty_like(emissions)
        K.crf_grad(emissions, tags, mask, mod.start_transitions, mod.end_transitions, mod.transitions,
                   gllh.to(F32).contiguous(), de, A.g(mod.start_transitions), A.g(mod.end_transitions),
                   A.g(mod.transitions))
        A.flush_final()
        return None, de, None, None, None, None


class CRF(ArenaModule):
    def __init__(self, num_tags: int, batch_first: bool = False) -> None:
        if num_tags <= 0:
            raise ValueError("invalid number of tags: %d" % num_tags)
        if num_tags > 64:
            raise ValueError("icka_amd.CRF supports at most 64 tags (one lane per tag)")
        super().__init__()
        self.num_tags = num_tags
        self.batch_first = batch_first
        self.start_transitions = nn.Parameter(torch.empty(num_tags))
        self.end_transitions = nn.Parameter(torch.empty(num_tags))
        self.transitions = nn.Parameter(torch.empty(num_tags, num_tags))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        nn.init.uniform_(self.start_transitions, -0.1, 0.1)
        nn.init.uniform_(self.end_transitions, -0.1, 0.1)
        nn.init.uniform_(self.transitions, -0.1, 0.1)

    def __repr__(self) -> str:
        return "%s(num_tags=%d)" % (self.__class__.__name__, self.num_tags)

    # ------------------------------------------------------------------------------------------------------------
    def _prep(self, emissions, tags, mask):
        if emissions.dim() != 3:
            raise ValueError("emissions must have dimension of 3, got %d" % emissions.dim())
        if emissions.size(2) != self.num_tags:
            raise ValueError("expected last dimension of emissions is %d, got %d" % (self.num_tags, emissions.size(2)))
        if not emissions.is_cuda:
            raise TypeError("icka_amd.CRF: emissions are on %s; there is no CPU path" % emissions.device)
        if not self.batch_first:   # the kernels are batch-first
            emissions = emissions.transpose(0, 1)
            tags = tags.transpose(0, 1) if tags is not None else None
            mask = mask.transpose(0, 1) if mask is not None else None
        B, S, _ = emissions.shape
        if tags is not None and tuple(tags.shape) != (B, S):
            raise ValueError("the first two dimensions of emissions and tags must match, got %s and %s"
                             % ((B, S), tuple(tags.shape)))
        if mask is not None:
            if tuple(mask.shape) != (B, S):
                raise ValueError("the first two dimensions of emissions and mask must match, got %s and %s"
                                 % ((B, S), tuple(mask.shape)))
            mask = (mask != 0).to(torch.int64).contiguous()
        e = emissions.to(F32).contiguous()
        t = tags.to(torch.int64).contiguous() if tags is not None else None
        return e, t, mask

    def forward(self, emissions: torch.Tensor, tags: torch.Tensor, mask: Optional[torch.Tensor] = None,
                reduction: str = "sum") -> torch.Tensor:
        """Log-likelihood of ``tags`` given ``emissions`` (``none`` | ``sum`` | ``mean`` | ``token_mean``).
        Like the package, the first position of every sequence is treated as unmasked (the package raises when it is
        not; checking would cost a device->host sync, so it is not checked here)."""
        if reduction not in ("none", "sum", "mean", "token_mean"):
            raise ValueError("invalid reduction: %s" % reduction)
        e, t, m = self._prep(emissions, tags, mask)
        A = self._arena()
        llh = _CrfFn.apply(A.anchor, e, t, m, self, A)
        if reduction == "none":
            return llh
        if reduction == "sum":
            return llh.sum()
        if reduction == "mean":
            return llh.mean()
        n = m.to(F32).sum() if m is not None else float(e.shape[0] * e.shape[1])
        return llh.sum() / n

    def decode(self, emissions: torch.Tensor, mask: Optional[torch.Tensor] = None) -> List[List[int]]:
        """Most likely tag sequence per sample (Viterbi), as python lists of length sum(mask)."""
        e, _, m = self._prep(emissions.detach(), None, mask)
        self._arena()
        best = torch.empty(e.shape[0], e.shape[1], dtype=torch.int64, device=e.device)
        K.crf_decode(e, m, self.start_transitions.detach(), self.end_transitions.detach(), self.transitions.detach(),
                     best)
        rows = best.cpu().tolist()
        return [[v for v in r if v >= 0] for r in rows]
