"""Bidirectional LSTM layer on the HIP kernels.

Drop-in for the ``nn.LSTM(input_size=H, hidden_size=H, batch_first=True, bidirectional=True)`` the reference puts in
front of its classifier (Cross_Modal_Interaction_Module.py:905-908, called ``x, _ = self.lstm(result)`` at :1042):
same parameter names (``weight_ih_l0`` ... ``bias_hh_l0_reverse``, so the reference's state_dict loads), same
initialisation, same ``(output, (h_n, c_n))`` return.  The input projection, the weight gradients and the input
gradient are GEMMs of the GEMM kernels over all time steps; the recurrence is `icka_lstm_fwd/bwd` (one launch per
step, both directions per launch)."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import kernels as K
from .arena import ArenaModule, ParamArena, arena_of

BF16, F32 = torch.bfloat16, torch.float32


class _LstmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, mod, A: ParamArena, B: int, S: int):
        H = mod.hidden_size
        M = B * S
        dev = x.device
        gx = torch.empty(M, 8 * H, dtype=F32, device=dev)
        K.gemm(K.GEMM_NT, x, A.w_cat((mod.weight_ih_l0, mod.weight_ih_l0_reverse)), gx,
               bias=A.f_cat((mod.bias_ih_l0, mod.bias_ih_l0_reverse)),
               bias2=A.f_cat((mod.bias_hh_l0, mod.bias_hh_l0_reverse)))
        y = torch.empty(M, 2 * H, dtype=BF16, device=dev)
        c_all = torch.empty(M, 2 * H, dtype=F32, device=dev)
        act = torch.empty(M, 8 * H, dtype=BF16, device=dev)
        save = any(ctx.needs_input_grad)
        hprev = torch.empty(M, 2 * H, dtype=BF16, device=dev) if save else None
        K.lstm_fwd(gx, A.w_cat((mod.weight_hh_l0, mod.weight_hh_l0_reverse)), y, c_all, act, hprev, B, S, H,
                   flags=mod.recurrence_flags)
        ctx.mod, ctx.A, ctx.dims = mod, A, (B, S, H)
        ctx.need_dx = x.requires_grad
        if save:
            ctx.save_for_backward(x, c_all, act, hprev)
        ctx.mark_non_differentiable(c_all)
        return y, c_all

    @staticmethod
    def backward(ctx, dy, _dc):
        x, c_all, act, hprev = ctx.saved_tensors
        mod, A = ctx.mod, ctx.A
        B, S, H = ctx.dims
        M = B * S
        dev = x.device
        if dy.dtype != BF16:
            raise TypeError("LSTM output gradient must be bf16, got %s" % dy.dtype)
        dy = dy if dy.is_contiguous() else dy.contiguous()
        whh = A.w_cat((mod.weight_hh_l0, mod.weight_hh_l0_reverse))           # [2][4H][H]
        whh_t = K.transpose_bf16(whh, torch.empty(2 * H, 4 * H, dtype=BF16, device=dev), 2, 4 * H, H)
        dgates = torch.empty(M, 8 * H, dtype=BF16, device=dev)
        carry = torch.empty(2 * B, H, dtype=F32, device=dev)
        K.lstm_bwd(dy, whh_t, act, c_all, dgates, carry, B, S, H, flags=mod.recurrence_flags)
        # parameter gradients: GEMMs / column sums over all steps
        wih = (mod.weight_ih_l0, mod.weight_ih_l0_reverse)
        K.gemm(K.GEMM_TN, dgates, x, A.g_cat(wih), beta=A.grad_beta(wih))
        for d, w in enumerate((mod.weight_hh_l0, mod.weight_hh_l0_reverse)):
            K.gemm(K.GEMM_TN, dgates[:, d * 4 * H:(d + 1) * 4 * H], hprev[:, d * H:(d + 1) * H], A.g(w),
                   beta=A.grad_beta(w))
        csw = A.workspace("colsum", K._lib.load().icka_colsum_workspace_floats(8 * H))
        for bs in ((mod.bias_ih_l0, mod.bias_ih_l0_reverse), (mod.bias_hh_l0, mod.bias_hh_l0_reverse)):
            K.colsum(dgates, A.g_cat(bs), csw, accumulate=A.grad_beta(bs) > 0)
        dx = None
        if ctx.need_dx:
            dx = torch.empty_like(x)
            K.gemm(K.GEMM_NN, dgates, A.w_cat(wih), dx)
        A.flush_final()
        return None, dx, None, None, None, None


class BiLSTM(ArenaModule):
    """nn.LSTM(input_size, hidden_size, num_layers=1, batch_first=True, bidirectional=True) on MI355X."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1, bias: bool = True,
                 batch_first: bool = True, dropout: float = 0.0, bidirectional: bool = True):
        super().__init__()
        if num_layers != 1 or not bias or not bidirectional or dropout != 0.0:
            raise ValueError("icka_amd.BiLSTM implements the reference's configuration: one bidirectional layer with "
                             "biases and no dropout (Cross_Modal_Interaction_Module.py:905-908)")
        if hidden_size % 32 or input_size % 8:
            raise ValueError("hidden_size must be a multiple of 32 and input_size of 8 (MFMA / 16-byte tiles)")
        self.config = None
        self.input_size, self.hidden_size, self.batch_first = input_size, hidden_size, batch_first
        # which form of the recurrence this module's calls take (icka_hip.h: ICKA_LSTM_*; 0 = one persistent launch with the
        # tagged-word hand-off).  A per-module, per-call argument: the library keeps no process-wide switch.
        self.recurrence_flags = 0
        H = hidden_size
        for sfx in ("", "_reverse"):
            self.register_parameter("weight_ih_l0" + sfx, nn.Parameter(torch.empty(4 * H, input_size)))
            self.register_parameter("weight_hh_l0" + sfx, nn.Parameter(torch.empty(4 * H, H)))
            self.register_parameter("bias_ih_l0" + sfx, nn.Parameter(torch.empty(4 * H)))
            self.register_parameter("bias_hh_l0" + sfx, nn.Parameter(torch.empty(4 * H)))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        stdv = 1.0 / math.sqrt(self.hidden_size)
        for p in self.parameters():
            nn.init.uniform_(p, -stdv, stdv)

    def icka_param_order(self):
        # both directions of each tensor back to back: [8H, in] / [8H, H] fused operands for the GEMMs and the kernel
        names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")
        return [(n + s, getattr(self, n + s)) for n in names for s in ("", "_reverse")]

    def forward(self, x: torch.Tensor, hx=None):
        if hx is not None:
            raise NotImplementedError("initial states other than zero are not used by the reference")
        if not x.is_cuda:
            raise TypeError("BiLSTM input must be on a ROCm device: icka_amd has no CPU path")
        if x.dim() != 3 or x.shape[-1] != self.input_size:
            raise ValueError("expected [batch, seq, %d] input" % self.input_size)
        if not self.batch_first:
            x = x.transpose(0, 1)
        B, S, _ = x.shape
        if B > 64:
            raise ValueError("at most 64 sequences per call")
        from .modeling import _hidden2d, _is_exact
        A = self._arena()
        H = self.hidden_size
        # host touch-point: a failed hand-off of an EARLIER persistent launch (its outputs are NaN-poisoned) is raised here;
        # the check is a host read of a mapped word -- no device synchronisation, legal inside a stream capture
        K.lstm_check_error("detected at the next BiLSTM.forward")
        if _is_exact(self):     # fp32 mode: f32 in / out, per-step f32 GEMMs (icka_amd/exact.py)
            from . import exact as X
            y, c_all = X.LstmFn.apply(A.anchor, _hidden2d(x, "input", True), self, A, B, S)
        else:
            y, c_all = _LstmFn.apply(A.anchor, _hidden2d(x, "input"), self, A, B, S)
        out = y.view(B, S, 2 * H)
        # h_n / c_n as nn.LSTM: [2, B, H] -- forward direction's last step, reverse direction's step at t = 0
        c4 = c_all.view(B, S, 2, H)
        h_n = torch.stack((out[:, S - 1, :H], out[:, 0, H:]), 0).float()
        c_n = torch.stack((c4[:, S - 1, 0], c4[:, 0, 1]), 0)
        if not self.batch_first:
            out = out.transpose(0, 1)
        return out, (h_n, c_n)
