"""Drop-in nn.Modules for the ICKA MNER hot path, running on hand-written gfx950 kernels.

Same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys as the reference's blocks
(Cross_Modal_Interaction_Module.py:302-698, my_bert/gate_cl_modeling.py:239-604, a_transformers/modeling_bert.py
encoder-only branch) and its gated MNER head (my_bert/cl_modeling.py:1252-1388, gate_cl_modeling.py:1248-1400), so
checkpoints load unchanged and callers such as My_cross_attention.py:676-683/:814-817 can switch imports.

Parameters are ordinary fp32 ``nn.Parameter``s (views into a ParamArena once the module has run on a ROCm device);
compute is bf16 MFMA with fp32 accumulation / statistics.  Hidden states are exchanged as bf16 ``[B,S,H]`` tensors
(fp32 inputs are cast on entry).  There is no CPU path: calling ``forward`` with CPU tensors raises.
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import torch
import torch.nn as nn

from . import exact as X
from . import kernels as K
from . import ops
from .arena import ArenaModule, ParamArena, arena_of, refresh_shadow_once
from .config import BertConfig, check_config, check_head_size_bf16

BF16, F32 = torch.bfloat16, torch.float32


# ------------------------------------------------------------------------------------------------- small helpers
class _CastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, to_bf16: bool):
        ctx.to_bf16 = to_bf16
        x = x.contiguous()
        if to_bf16:
            return K.cast_f32_to_bf16(x, torch.empty_like(x, dtype=BF16))
        return K.cast_bf16_to_f32(x, torch.empty_like(x, dtype=F32))

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if ctx.to_bf16:
            return K.cast_bf16_to_f32(dy, torch.empty_like(dy, dtype=F32)), None
        return K.cast_f32_to_bf16(dy, torch.empty_like(dy, dtype=BF16)), None


def _hidden2d(x: torch.Tensor, name: str = "hidden_states", exact: bool = False) -> torch.Tensor:
    """[B,S,H] (bf16 or f32, ROCm) -> contiguous bf16 [B*S, H]  (f32 [B*S, H] in the fp32-exact mode)."""
    if not x.is_cuda:
        raise TypeError("%s must be on a ROCm device: icka_amd has no CPU path" % name)
    if exact:
        if x.dtype == BF16:
            x = _CastFn.apply(x, False)
        elif x.dtype != F32:
            raise TypeError("%s must be bf16 or f32, got %s" % (name, x.dtype))
        if not x.is_contiguous():
            x = x.contiguous()
        return x.view(-1, x.shape[-1])
    if x.dtype == F32:
        x = _CastFn.apply(x, True)
    elif x.dtype != BF16:
        raise TypeError("%s must be bf16 or f32, got %s" % (name, x.dtype))
    if not x.is_contiguous():
        x = x.contiguous()
    return x.view(-1, x.shape[-1])


def _twin(x: torch.Tensor) -> Optional[torch.Tensor]:
    """f32 twin of a bf16 hidden-state tensor produced by one of our blocks (carried as a Python attribute so the
    public signatures stay the reference's): the residual stream is accumulated in f32."""
    t = getattr(x, "_icka_f32", None)
    if t is None or t.numel() != x.numel() or t.device != x.device:
        return None
    return t.view(-1, x.shape[-1])


def _with_twin(y2d: torch.Tensor, yf2d: Optional[torch.Tensor], shape) -> torch.Tensor:
    out = y2d.view(shape)
    if yf2d is not None:
        out._icka_f32 = yf2d
    return out


def _add_mask2d(mask: torch.Tensor, B: int, T: int) -> torch.Tensor:
    """The reference hands blocks the *extended additive* mask [B,1,1,T] ((1-m)*-10000, :364-372): flatten to the
    f32 [B,T] the attention kernel reads."""
    if mask.dtype != F32:
        mask = mask.float()
    if mask.numel() != B * T:
        mask = mask.expand(B, 1, 1, T)
    return mask.reshape(B, T).contiguous()


PRECISIONS = ("auto", "bf16", "mixed16", "fp32")
AUTO_MIXED16_DEPTH = 12     # "auto": stacks deeper than this run mixed16 (pure bf16 exceeds the 2e-2 logit bar at 24 layers)


def set_precision(module: nn.Module, precision: str) -> nn.Module:
    """Select the arithmetic of every icka block under ``module``:
    "auto" (the default of a module nobody called set_precision on): "mixed16" for stacks of more than 12 encoder layers
               (``config.num_hidden_layers``; bert-large, BASELINE config c4, measures 2.2e-2 .. 2.5e-2 in pure bf16,
               above north_star's 2e-2, and 3.8e-3 in mixed16), "bf16" otherwise;
    "bf16": bf16 MFMA operands, f32 accumulation / statistics / residual stream -- the product path up to 12 layers;
    "mixed16": every FORWARD GEMM of the path reads IEEE fp16 operands (activations: an fp16 copy that is also the residual
               stream; weights: an fp16 shadow of the masters) on v_mfma_f32_16x16x32_f16 -- 11 significand bits instead
               of 8, same MFMA rate: the encoder and cross layers (q/k/v, out-proj, FFN), the region projection (fp16 region
               tokens made from the f32 features), the K/V projections of the cross layers (fp16 twin of the projected
               regions), the relevance-scaled cross stream of the gate_cl head, the gate GEMM and the classifier.  The
               OUTPUTS q / k / v (the attention kernels' operand type), the attention kernels and the WHOLE backward stay
               bf16 (no loss scaling needed: no gradient is ever held in fp16).  For deep stacks whose bf16 rounding noise
               exceeds the 2e-2 logit bar (bert-large, BASELINE config c4).  It is implemented by the fused layer Functions
               (BertEmbeddings, BertLayer, BertCrossAttentionLayer and the models built from them); sub-modules called one
               by one (BertAttention, BertIntermediate, ...) and the hf_style shim keep bf16 operands;
    "fp32": f32 storage and f32-input MFMA arithmetic (icka_amd/exact.py) for the 1e-3 parity bar of BASELINE.json."""
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (PRECISIONS,))
    for m in module.modules():
        m.icka_precision = precision
    return module


def resolved_precision(module: nn.Module) -> str:
    """The arithmetic mode ``module`` runs in: its ``icka_precision`` ("auto" when set_precision was never called), with
    "auto" resolved from the depth of the stack the module belongs to (its ``config.num_hidden_layers``)."""
    p = getattr(module, "icka_precision", "auto")
    if p != "auto":
        return p
    depth = getattr(getattr(module, "config", None), "num_hidden_layers", 0) or 0
    return "mixed16" if depth > AUTO_MIXED16_DEPTH else "bf16"


def _is_exact(module: nn.Module) -> bool:
    return getattr(module, "icka_precision", "auto") == "fp32"


def _is_mixed(module: nn.Module) -> bool:
    return resolved_precision(module) == "mixed16"


def _dims(config, B, S, R, train: bool, exact: bool = False, mixed: bool = False) -> ops.Dims:
    if not exact:
        check_head_size_bf16(config)
    return ops.Dims(B, S, R, config.hidden_size, config.intermediate_size, config.num_attention_heads,
                    float(getattr(config, "layer_norm_eps", 1e-12)), float(config.hidden_dropout_prob),
                    float(config.attention_probs_dropout_prob), train, h16=mixed and not exact)


class _IckaModule(ArenaModule):
    """Shared plumbing (arena.ArenaModule): lazily (re)build the ParamArena over the outermost module; the outermost
    forward of a call tree refreshes the bf16 shadows."""

    def _anchor(self, A: ParamArena) -> torch.Tensor:
        return A.anchor


# ------------------------------------------------------------------------------------------------- blocks
class BertLayerNorm(_IckaModule):
    """TF-style LayerNorm, eps inside the sqrt (Cross_Modal_Interaction_Module.py:509-522)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        A = self._arena()
        shape = x.shape
        if _is_exact(self):
            return X.LayerNormFn.apply(A.anchor, _hidden2d(x, "x", True), None, self, A,
                                       float(self.variance_epsilon)).view(shape)
        return _LayerNormFn.apply(A.anchor, _hidden2d(x, "x"), self, A).view(shape)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, mod, A):
        M, H = x.shape
        y = torch.empty_like(x)
        xhat = torch.empty_like(x)
        rstd = torch.empty(M, dtype=F32, device=x.device)
        K.ln_fwd(x, None, None, mod.weight, mod.bias, y, xhat=xhat, rstd=rstd, eps=mod.variance_epsilon)
        ctx.mod, ctx.A = mod, A
        ctx.save_for_backward(xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xhat, rstd = ctx.saved_tensors
        mod, A = ctx.mod, ctx.A
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        ws = A.workspace("ln", K._lib.load().icka_ln_bwd_workspace_floats(dy.shape[1]))
        acc = A.grad_beta((mod.weight, mod.bias)) > 0
        K.ln_bwd(dy, xhat, rstd, mod.weight, dres=dx, dgamma=A.g(mod.weight), dbeta=A.g(mod.bias), partials=ws,
                 accumulate=acc)
        A.flush_final()
        return None, dx, None, None


class BertEmbeddings(_IckaModule):
    """word + position + token_type -> LayerNorm -> dropout (:384-412)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-12))
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None):
        if not input_ids.is_cuda:
            raise TypeError("input_ids must be on a ROCm device: icka_amd has no CPU path")
        B, S = input_ids.shape
        if S > self.position_embeddings.weight.shape[0]:
            raise IndexError("sequence length %d exceeds max_position_embeddings" % S)
        A = self._arena()
        ids = input_ids.contiguous()
        tt = None if token_type_ids is None else token_type_ids.contiguous()
        if _is_exact(self):
            d = _dims(self.config, B, S, 0, self.training, True)
            return X.EmbeddingsFn.apply(A.anchor, self, A, ids, tt, d).view(B, S, -1)
        d = _dims(self.config, B, S, 0, self.training, mixed=_is_mixed(self))
        y, yf = ops.EmbeddingsFn.apply(A.anchor, self, A, ids, tt, d)
        return _with_twin(y, yf, (B, S, -1))


class BertSelfAttention(_IckaModule):
    """forward(hidden_states, attention_mask) -> context_layer [B,S,H] (:456-506).  Inside BertLayer the same kernels
    run with the block's gradient fan-ins fused; called on its own (as BertAttention.forward does, :451-454) the
    projections + fused attention run as one autograd node."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.config = config
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)

    def icka_param_order(self):
        # weights back to back, then biases: one [3H,H] (or [2H,H] K|V) operand for the fused projection GEMM
        return [("query.weight", self.query.weight), ("key.weight", self.key.weight),
                ("value.weight", self.value.weight), ("query.bias", self.query.bias),
                ("key.bias", self.key.bias), ("value.bias", self.value.bias)]

    def _context(self, q_states, kv_states, attention_mask):
        B, S, H = q_states.shape
        A = self._arena()
        ex = _is_exact(self)
        x = _hidden2d(q_states, "hidden_states", ex)
        kv = None if kv_states is None else _hidden2d(kv_states, "s2_hidden_states", ex)
        T = S if kv_states is None else kv_states.shape[1]
        d = _dims(self.config, B, S, T if kv_states is not None else 0, self.training, ex)
        mask = _add_mask2d(attention_mask, B, T)
        fn = X.AttnCoreFn if ex else ops.AttnCoreFn
        return fn.apply(A.anchor, x, kv, self, A, mask, d, T).view(B, S, H)

    def forward(self, hidden_states, attention_mask):
        return self._context(hidden_states, None, attention_mask)


class BertCoAttention(BertSelfAttention):
    """forward(s1_hidden_states, s2_hidden_states, s2_attention_mask): Q from s1, K/V from s2 (:568-624).
    ``fp8_scores = True`` (BASELINE config c5, not a reference feature) computes QK^T and PV of this co-attention on the
    fp8 matrix cores; requires s1/s2 lengths <= 128."""
    fp8_scores = False

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask):
        return self._context(s1_hidden_states, s2_hidden_states, s2_attention_mask)


class _DenseResidualNorm(_IckaModule):
    """LayerNorm(dropout(dense(hidden_states)) + input_tensor): BertSelfOutput (:554-565) and BertOutput (:525-536)."""

    def forward(self, hidden_states, input_tensor):
        shape = input_tensor.shape
        A = self._arena()
        ex = _is_exact(self)
        h = _hidden2d(hidden_states, "hidden_states", ex)
        res = _hidden2d(input_tensor, "input_tensor", ex)
        B = h.shape[0]
        d = _dims(self.config, B, 1, 0, self.training, True)     # no attention here: head size is irrelevant
        if ex:
            return X.DenseResidualNormFn.apply(A.anchor, h, res, self, A, d).view(shape)
        y, yf = ops.DenseResidualNormFn.apply(A.anchor, h, res, _twin(input_tensor), self, A, d)
        return _with_twin(y, yf, shape)


class BertSelfOutput(_DenseResidualNorm):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-12))
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(_IckaModule):
    """forward(input_tensor, attention_mask) = output(self(input_tensor, attention_mask), input_tensor) (:445-454)."""

    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, input_tensor, attention_mask):
        self_output = self.self(input_tensor, attention_mask)
        return self.output(self_output, input_tensor)


class BertCrossAttention(_IckaModule):
    """forward(s1_input_tensor, s2_input_tensor, s2_attention_mask) (:627-636)."""

    def __init__(self, config):
        super().__init__()
        self.self = BertCoAttention(config)
        self.output = BertSelfOutput(config)

    def forward(self, s1_input_tensor, s2_input_tensor, s2_attention_mask):
        s1_cross_output = self.self(s1_input_tensor, s2_input_tensor, s2_attention_mask)
        return self.output(s1_cross_output, s1_input_tensor)


class BertIntermediate(_IckaModule):
    """forward(hidden_states) = gelu(dense(hidden_states)) (:539-551)."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        shape = hidden_states.shape
        A = self._arena()
        ex = _is_exact(self)
        x = _hidden2d(hidden_states, "hidden_states", ex)
        fn = X.IntermediateFn if ex else ops.IntermediateFn
        return fn.apply(A.anchor, x, self, A).view(*shape[:-1], -1)


class BertOutput(_DenseResidualNorm):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=getattr(config, "layer_norm_eps", 1e-12))
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(_IckaModule):
    """forward(hidden_states, attention_mask) with the extended additive mask, as :431-442."""

    def __init__(self, config):
        super().__init__()
        check_config(config)
        self.config = config
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, attention_mask):
        B, S, H = hidden_states.shape
        A = self._arena()
        if _is_exact(self):
            d = _dims(self.config, B, S, 0, self.training, True)
            return X.BertLayerFn.apply(A.anchor, _hidden2d(hidden_states, exact=True), self, A,
                                       _add_mask2d(attention_mask, B, S), d).view(B, S, H)
        x = _hidden2d(hidden_states)
        d = _dims(self.config, B, S, 0, self.training, mixed=_is_mixed(self))
        y, yf = ops.BertLayerFn.apply(A.anchor, x, _twin(hidden_states), self, A, _add_mask2d(attention_mask, B, S), d)
        return _with_twin(y, yf, (B, S, H))


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        layer = BertLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask, output_all_encoded_layers=True):
        all_encoder_layers = []
        for layer_module in self.layer:
            hidden_states = layer_module(hidden_states, attention_mask)
            if output_all_encoded_layers:
                all_encoder_layers.append(hidden_states)
        if not output_all_encoded_layers:
            all_encoder_layers.append(hidden_states)
        return all_encoder_layers


class BertSelfEncoder(nn.Module):
    """One-layer encoder (:683-698); constructed by the reference heads, unused on the hot path."""

    def __init__(self, config):
        super().__init__()
        layer = BertLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(1)])

    def forward(self, hidden_states, attention_mask, output_all_encoded_layers=True):
        return BertEncoder.forward(self, hidden_states, attention_mask, output_all_encoded_layers)


class BertCrossAttentionLayer(_IckaModule):
    """forward(s1_hidden_states, s2_hidden_states, s2_attention_mask) (:639-650)."""

    def __init__(self, config):
        super().__init__()
        check_config(config)
        self.config = config
        self.attention = BertCrossAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask):
        B, S, H = s1_hidden_states.shape
        R = s2_hidden_states.shape[1]
        A = self._arena()
        if _is_exact(self):
            d = _dims(self.config, B, S, R, self.training, True)
            return X.CrossLayerFn.apply(A.anchor, _hidden2d(s1_hidden_states, "s1_hidden_states", True),
                                        _hidden2d(s2_hidden_states, "s2_hidden_states", True), self, A,
                                        _add_mask2d(s2_attention_mask, B, R), d).view(B, S, H)
        s1 = _hidden2d(s1_hidden_states, "s1_hidden_states")
        s2 = _hidden2d(s2_hidden_states, "s2_hidden_states")
        d = _dims(self.config, B, S, R, self.training, mixed=_is_mixed(self))
        y, yf = ops.CrossLayerFn.apply(A.anchor, s1, _twin(s1_hidden_states), s2, self, A,
                                       _add_mask2d(s2_attention_mask, B, R), d)
        return _with_twin(y, yf, (B, S, H))


class BertCrossEncoder(nn.Module):
    """forward(s1, s2, s2_attention_mask, output_all_encoded_layers=True) -> list (:653-667)."""

    def __init__(self, config, layer_num):
        super().__init__()
        layer = BertCrossAttentionLayer(config)
        self.layer = nn.ModuleList([copy.deepcopy(layer) for _ in range(layer_num)])

    def forward(self, s1_hidden_states, s2_hidden_states, s2_attention_mask, output_all_encoded_layers=True):
        all_encoder_layers = []
        for layer_module in self.layer:
            s1_hidden_states = layer_module(s1_hidden_states, s2_hidden_states, s2_attention_mask)
            if output_all_encoded_layers:
                all_encoder_layers.append(s1_hidden_states)
        if not output_all_encoded_layers:
            all_encoder_layers.append(s1_hidden_states)
        return all_encoder_layers


class BertPooler(_IckaModule):
    """tanh(dense(hidden_states[:, 0])) (:669-681)."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        A = self._arena()
        B, S, H = hidden_states.shape
        if _is_exact(self):
            first = _hidden2d(hidden_states, exact=True).view(B, S, H)[:, 0]
            return X.LinearFn.apply(A.anchor, first, self.dense, A, True)
        x = _hidden2d(hidden_states)
        first = x.view(B, S, H)[:, 0]            # strided [B,H] view, row stride S*H: read in place by the GEMM
        return ops.LinearFn.apply(A.anchor, first, self.dense, A, False, K.EPI_TANH)


class cls_layer_both(_IckaModule):
    """proj(LayerNorm(lang_feat + img_feat)) (Cross_Modal_Interaction_Module.py:873-884; ``proj_norm`` and
    ``LayerNorm`` are one nn.LayerNorm, eps 1e-5, exactly as the reference aliases them)."""

    def __init__(self, input_dim, output_dim):
        super().__init__()
        self.proj_norm = self.LayerNorm = nn.LayerNorm(input_dim)
        self.proj = nn.Linear(input_dim, output_dim)

    def forward(self, lang_feat, img_feat):
        A = self._arena()
        if _is_exact(self):
            x = _hidden2d(lang_feat, "lang_feat", True)
            r = _hidden2d(img_feat, "img_feat", True)
            feat = X.LayerNormFn.apply(A.anchor, x, r, self.proj_norm, A, float(self.proj_norm.eps))
            return X.LinearFn.apply(A.anchor, feat, self.proj, A, False)
        x = lang_feat if lang_feat.dtype == BF16 else _CastFn.apply(lang_feat, True)
        r = img_feat if img_feat.dtype == BF16 else _CastFn.apply(img_feat, True)
        feat = ops.AddLayerNormFn.apply(A.anchor, x, r, self.proj_norm, A, float(self.proj_norm.eps))
        return ops.LinearFn.apply(A.anchor, feat, self.proj, A, False, K.EPI_NONE)


def scalar_gate_fusion(owner: nn.Module, cross_output_layer: torch.Tensor, token_embedding: torch.Tensor):
    """Lines 1029-1036 of the reference's current model: ``owner`` exposes ``cls_layer`` (cls_layer_both) and
    ``aux_head`` (nn.Linear(H,1));  g = sigmoid(aux_head(cls_layer(cross[:,0], tok[:,0])));
    returns g*token_embedding + (1-g)*cross_output_layer  ([B,S,H] bf16).  ``token_embedding`` comes from the
    out-of-scope RoBERTa stage and is an input here."""
    A = arena_of(owner)
    refresh_shadow_once(A)      # once per outermost forward (every call when used as a free function)
    B, S, H = cross_output_layer.shape
    if _is_exact(owner):
        cross = _hidden2d(cross_output_layer, "cross_output_layer", True)
        tok = _hidden2d(token_embedding, "token_embedding", True)
        feat = X.LayerNormFn.apply(A.anchor, cross.view(B, S, H)[:, 0], tok.view(B, S, H)[:, 0], owner.cls_layer.proj_norm,
                                   A, float(owner.cls_layer.proj_norm.eps))
        related = X.LinearFn.apply(A.anchor, feat, owner.cls_layer.proj, A, False)
        logit = X.LinearFn.apply(A.anchor, related, owner.aux_head, A, False)
        return X.SampleGateFn.apply(tok, cross, logit.view(B), 0, B, S).view(B, S, H)
    cross = _hidden2d(cross_output_layer, "cross_output_layer")
    tok = _hidden2d(token_embedding, "token_embedding")
    c0 = cross.view(B, S, H)[:, 0]          # strided [B,H] views (row stride S*H), read in place
    t0 = tok.view(B, S, H)[:, 0]
    feat = ops.AddLayerNormFn.apply(A.anchor, c0, t0, owner.cls_layer.proj_norm, A, float(owner.cls_layer.proj_norm.eps))
    related = ops.LinearFn.apply(A.anchor, feat, owner.cls_layer.proj, A, False, K.EPI_NONE)
    logit = ops.LinearFn.apply(A.anchor, related, owner.aux_head, A, True, K.EPI_NONE)        # f32 [B,1]
    out = ops.SampleGateFn.apply(tok, cross, logit.view(B), 0, B, S)
    return out.view(B, S, H)


# ------------------------------------------------------------------------------------------------- models
class BertPreTrainedModel(_IckaModule):
    """Weight init + checkpoint key handling of the reference (:141-299).  ``from_pretrained`` takes a directory
    holding ``bert_config.json``/``config.json`` + ``pytorch_model.bin`` or an explicit ``state_dict=``; the
    reference's S3 download / TF-checkpoint paths are I/O outside the hot path and are not reproduced."""

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        for f in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                  "hidden_dropout_prob", "attention_probs_dropout_prob", "max_position_embeddings",
                  "type_vocab_size", "vocab_size"):
            if not hasattr(config, f):
                raise ValueError("Parameter config in `{}(config)` should expose BertConfig fields (missing `{}`)"
                                 .format(self.__class__.__name__, f))
        self.config = config

    def init_bert_weights(self, module):
        std = getattr(self.config, "initializer_range", 0.02)
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=std)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @staticmethod
    def convert_legacy_keys(state_dict):
        """gamma/beta -> weight/bias renaming of old checkpoints (:256-268)."""
        out = type(state_dict)()
        for key, v in state_dict.items():
            nk = key
            if "gamma" in nk:
                nk = nk.replace("gamma", "weight")
            if "beta" in nk:
                nk = nk.replace("beta", "bias")
            out[nk] = v
        return out

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *inputs, **kwargs):
        import json
        import os
        state_dict = kwargs.pop("state_dict", None)
        kwargs.pop("cache_dir", None)
        if kwargs.pop("from_tf", False):
            raise NotImplementedError("TensorFlow checkpoints: convert to a PyTorch state_dict first")
        config = kwargs.pop("config", None)
        if config is None:
            for n in ("config.json", "bert_config.json"):
                p = os.path.join(pretrained_model_name_or_path, n)
                if os.path.exists(p):
                    config = BertConfig.from_json_file(p)
                    break
        if config is None:
            raise EnvironmentError("no config.json / bert_config.json under %r" % (pretrained_model_name_or_path,))
        model = cls(config, *inputs, **kwargs)
        if state_dict is None:
            state_dict = torch.load(os.path.join(pretrained_model_name_or_path, "pytorch_model.bin"),
                                    map_location="cpu")
        state_dict = cls.convert_legacy_keys(state_dict)
        # a bare BertModel loads 'bert.'-prefixed checkpoints (:286-289)
        if not hasattr(model, "bert") and any(k.startswith("bert.") for k in state_dict):
            state_dict = type(state_dict)((k[5:], v) for k, v in state_dict.items() if k.startswith("bert."))
        missing, unexpected = model.load_state_dict(state_dict, strict=False)
        model._load_report = {"missing": list(missing), "unexpected": list(unexpected)}
        return model


class BertModel(BertPreTrainedModel):
    """forward(input_ids, token_type_ids=None, attention_mask=None, output_all_encoded_layers=True)
    -> (encoded_layers, pooled_output), as :302-382."""

    def __init__(self, config):
        super().__init__(config)
        check_config(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config)
        self.apply(self.init_bert_weights)

    def encode(self, input_ids, token_type_ids=None, attention_mask=None, output_all_encoded_layers=True):
        """forward() without the pooler: the list of encoded layers (:364-379).  The MNER heads read only the sequence
        output (cl_modeling.py:1341-1344), so their trunk does not launch the pooler GEMM for a result nobody uses."""
        if not input_ids.is_cuda:
            raise TypeError("input_ids must be on a ROCm device: icka_amd has no CPU path")
        B, S = input_ids.shape
        self._arena()
        add_mask = torch.zeros(B, S, dtype=F32, device=input_ids.device) if attention_mask is None else \
            K.additive_mask(attention_mask.long() if attention_mask.dtype != torch.int64 else attention_mask, S,
                            torch.empty(B, S, dtype=F32, device=input_ids.device))
        extended_attention_mask = add_mask.view(B, 1, 1, S)
        embedding_output = self.embeddings(input_ids, token_type_ids)
        return self.encoder(embedding_output, extended_attention_mask, output_all_encoded_layers=output_all_encoded_layers)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, output_all_encoded_layers=True):
        encoded_layers = self.encode(input_ids, token_type_ids, attention_mask, output_all_encoded_layers)
        sequence_output = encoded_layers[-1]
        pooled_output = self.pooler(sequence_output)
        if not output_all_encoded_layers:
            encoded_layers = encoded_layers[-1]
        return encoded_layers, pooled_output


def _mner_trunk(self, input_ids, segment_ids, input_mask, added_attention_mask, visual_embeds_att):
    """Shared trunk of the MNER heads: BERT text encoder -> dropout -> region tokens -> vismap2text -> text->image cross
    encoder (cl_modeling.py:1341-1361 = Cross_Modal_Interaction_Module.py:949-969 / :2446-2466).  ``self`` provides
    ``config, bert, vismap2text, txt2img_attention``.  Returns (arena, seq bf16 [B*S,H], its f32 twin or None,
    cross bf16 [B*S,H], its f32 twin)."""
    cfg = self.config
    B, S = input_ids.shape
    H = cfg.hidden_size
    A = self._arena()
    dev = input_ids.device
    # ---- text encoder (cl_modeling.py:1341-1344)
    if isinstance(self.bert, BertModel):
        sequence_output = self.bert.encode(input_ids, segment_ids, input_mask, output_all_encoded_layers=False)[-1]
    else:   # a caller-supplied text encoder with the reference's BertModel call convention
        sequence_output, _ = self.bert(input_ids, token_type_ids=segment_ids, attention_mask=input_mask,
                                       output_all_encoded_layers=False)
    exact = _is_exact(self)
    if exact:
        return _mner_trunk_exact(self, A, sequence_output.view(B * S, H), added_attention_mask, visual_embeds_att, B, S)
    seq = sequence_output.view(B * S, H)
    seqf = _twin(sequence_output)
    if self.training and cfg.hidden_dropout_prob > 0:
        seq = ops.DropoutFn.apply(seq, A, float(cfg.hidden_dropout_prob))
        seqf = None
    # ---- region tokens + projection (:1348-1350)
    v = visual_embeds_att
    R, layout = _region_layout(v)
    # region-token rows are padded to a multiple of 128 (zero rows) so that the region projection, the cross-attention K/V
    # projection and their gradients stay on the aligned GEMM kernels for every region count (the reference's own layout
    # has 49 regions: 32 x 49 = 1568 rows); the attention kernels address keys as b*R + r and never see the padding
    rows_p = (B * R + 127) // 128 * 128
    tokens = torch.empty(rows_p, 2048, dtype=BF16, device=dev)
    K.zero_rows_(tokens, B * R)
    vsrc = v.float().contiguous() if v.dtype != F32 or not v.is_contiguous() else v
    mixed = _is_mixed(self)
    vis16 = None
    if mixed:
        # mixed16: the region projection reads fp16 tokens (made from the f32 features in the same pass as the bf16 ones)
        # and fp16 weights, and hands the cross encoder an fp16 twin of the projected regions for its K/V projections
        tokens16 = torch.empty(rows_p, 2048, dtype=torch.float16, device=dev)
        K.zero_rows_(tokens16, B * R)
        K.regions_to_tokens_h(vsrc, tokens, tokens16, B, R, 2048, layout)
        vis, vis16 = ops.LinearFn.apply(A.anchor, tokens, self.vismap2text, A, False, K.EPI_NONE, tokens16)
    else:
        K.regions_to_tokens(vsrc, tokens, B, R, 2048, layout)
        vis = ops.LinearFn.apply(A.anchor, tokens, self.vismap2text, A, False, K.EPI_NONE)
    # ---- image mask (:1353-1356)
    img_mask = K.additive_mask(added_attention_mask if added_attention_mask.dtype == torch.int64
                               else added_attention_mask.long(), R, torch.empty(B, R, dtype=F32, device=dev))
    # ---- cross encoder (:1359-1361)
    # the text stream feeds both the cross encoder and the head: its two gradients are summed by our bf16 add kernel
    # (FanOutFn) instead of autograd's own accumulation
    if torch.is_grad_enabled() and seq.requires_grad:
        cross, seq = ops.FanOutFn.apply(seq, 2)
    else:
        cross = seq
    crossf = seqf
    for layer in self.txt2img_attention.layer:
        d = _dims(cfg, B, S, R, self.training, mixed=_is_mixed(layer))
        cross, crossf = ops.CrossLayerFn.apply(A.anchor, cross, crossf, vis, layer, A, img_mask, d,
                                               vis16 if d.h16 else None)
    return A, seq, seqf, cross, crossf


def _region_layout(v: torch.Tensor):
    """(R, layout): layout 1 = the reference's channel-major [B,2048,7,7] / [B,2048,R] (:956), 0 = tokens [B,R,2048]."""
    if not v.is_cuda:
        raise TypeError("visual_embeds_att must be on a ROCm device")
    if v.dim() == 4 or (v.dim() == 3 and v.shape[1] == 2048 and v.shape[2] != 2048):
        return (v.shape[2] * v.shape[3] if v.dim() == 4 else v.shape[2]), 1
    return v.shape[1], 0


def _mner_trunk_exact(self, A, seq, added_attention_mask, v, B, S):
    """fp32-exact form of the trunk after the text encoder (same reference lines as _mner_trunk)."""
    cfg = self.config
    dev = seq.device
    if self.training and cfg.hidden_dropout_prob > 0:
        seq = X.DropoutFn.apply(seq, A, float(cfg.hidden_dropout_prob))
    R, layout = _region_layout(v)
    tokens = X.regions_to_tokens(v if v.dtype == F32 else v.float(), B, R, layout)
    vis = X.LinearFn.apply(A.anchor, tokens, self.vismap2text, A, False)
    img_mask = K.additive_mask(added_attention_mask if added_attention_mask.dtype == torch.int64
                               else added_attention_mask.long(), R, torch.empty(B, R, dtype=F32, device=dev))
    d = _dims(cfg, B, S, R, self.training, True)
    cross = seq
    for layer in self.txt2img_attention.layer:
        cross = X.CrossLayerFn.apply(A.anchor, cross, vis, layer, A, img_mask, d)
    return A, seq, None, cross, None


class MTCCMBertForMMTokenClassificationCRF(BertPreTrainedModel):
    """Gated multimodal token classifier (my_bert/cl_modeling.py:1252-1388; gate_cl_modeling.py:1248-1400).

    forward(input_ids, segment_ids, input_mask, added_attention_mask, visual_embeds_mean, visual_embeds_att,
            temp=None, temp_lamb=None, lamb=None, labels=None, negative_rate=None)
    keeps the reference's positional signature (gate_cl_modeling.py:1319-1320).  The hot path ends at the per-token
    logits (``bert_feats``): with ``labels=None`` the logits ``[B,S,num_labels]`` (f32) are returned; with labels,
    the token-level cross-entropy over valid tokens (the benchmark loss).  The CRF and the
    contrastive / crs auxiliary losses of the reference are outside the hot path; a caller that owns a CRF module
    can assign it to ``self.crf`` and gets ``-crf(logits, labels, mask, reduction='mean')`` like the reference.

    ``visual_embeds_att`` is the myResnet 'att' tensor ``[B,2048,7,7]`` (R = 49, reference layout) or region
    tokens ``[B,R,2048]`` (BASELINE synthetic layout); ``regions`` defaults to 49 as in the reference.
    """

    def __init__(self, config, layer_num1=1, layer_num2=1, layer_num3=1, num_labels=2, regions=49, variant="cl",
                 max_seq_length=128, cross_attention_fp8=False, use_crf=False):
        super().__init__(config)
        check_config(config)
        if variant not in ("cl", "gate_cl"):
            raise ValueError("variant must be 'cl' (my_bert/cl_modeling.py) or 'gate_cl' (my_bert/gate_cl_modeling.py)")
        self.num_labels = num_labels
        self.regions = regions
        self.variant = variant
        self.bert = BertModel(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.vismap2text = nn.Linear(2048, config.hidden_size)
        self.txt2img_attention = BertCrossEncoder(config, layer_num1)
        if cross_attention_fp8:   # BASELINE config c5: fp8 QK^T / PV in the text->image co-attention
            for layer in self.txt2img_attention.layer:
                layer.attention.self.fp8_scores = True
        if variant == "gate_cl":
            # relevance score head: nn.Linear(hidden*2*128, 2) in the reference (gate_cl_modeling.py:1258); the
            # hard-coded 128 is the max_seq_length parameter here
            self.crs_classifier = nn.Linear(config.hidden_size * 2 * max_seq_length, 2)
        self.Gate_text = nn.Linear(config.hidden_size, config.hidden_size)
        self.Gate_image = nn.Linear(config.hidden_size, config.hidden_size)
        self.classifier = nn.Linear(config.hidden_size * 2, num_labels)
        # the reference builds torchcrf.CRF(num_labels, batch_first=True) here (cl_modeling.py:1269); use_crf=True builds
        # the HIP-backed equivalent (icka_amd.crf.CRF), any object with the same call/decode API may be assigned later
        self.crf = None
        if use_crf:
            from .crf import CRF
            self.crf = CRF(num_labels, batch_first=True)
        self.apply(self.init_bert_weights)

    def logits(self, input_ids, segment_ids, input_mask, added_attention_mask, visual_embeds_att):
        cfg = self.config
        B, S = input_ids.shape
        H = cfg.hidden_size
        A, seq, seqf, cross, crossf = _mner_trunk(self, input_ids, segment_ids, input_mask, added_attention_mask,
                                                  visual_embeds_att)
        if self.variant == "gate_cl":
            # P = softmax(crs_classifier(cat(seq, cross).view(B,-1)))[:, -1];  cross = P * cross   (:1364-1373)
            if self.crs_classifier.weight.shape[1] != 2 * H * S:
                raise ValueError("gate_cl: sequence length %d does not match crs_classifier (built for %d)"
                                 % (S, self.crs_classifier.weight.shape[1] // (2 * H)))
            if _is_exact(self):
                crs = X.CrsFn.apply(A.anchor, seq, cross, self.crs_classifier, A, B, S)
                cross = X.SampleGateFn.apply(cross, None, crs, 1, B, S)
            else:
                if torch.is_grad_enabled() and seq.requires_grad:
                    seq, seq_c = ops.FanOutFn.apply(seq, 2)
                    cross, cross_c = ops.FanOutFn.apply(cross, 2)
                else:
                    seq_c, cross_c = seq, cross
                crs = ops.CrsFn.apply(A.anchor, seq_c, cross_c, self.crs_classifier, A, B, S)
                if _is_mixed(self):
                    # mixed16: the relevance-scaled cross stream is formed from the fp16 twin of the cross encoder's output
                    # and leaves in both 16-bit types (the relevance score itself sums 2 H S products per sample: the bf16
                    # rounding of its operands averages out)
                    d16 = _dims(cfg, B, S, 0, self.training, mixed=True)
                    cross, crossf = ops.SampleGateFn.apply(cross, None, crs, 1, B, S, ops._fwd_twin(A, cross, crossf, d16))
                else:
                    cross = ops.SampleGateFn.apply(cross, None, crs, 1, B, S)
        # ---- gate + classifier (:1363-1371)
        if _is_exact(self):
            return X.GatedHeadFn.apply(A.anchor, seq, cross, self, A).view(B, S, self.num_labels)
        seq16 = cross16 = None
        if _is_mixed(self):
            # the fp16 twins of the two streams (left by the last encoder / cross layer / relevance gate; made from the bf16
            # tensors when a dropout or fan-out node sits in between) feed the gate GEMM and the classifier
            d16 = _dims(cfg, B, S, 0, self.training, mixed=True)
            seq16 = ops._fwd_twin(A, seq, seqf, d16)
            cross16 = ops._fwd_twin(A, cross, crossf, d16)
        logits = ops.GatedHeadFn.apply(A.anchor, seq, cross, self, A, seq16, cross16)
        return logits.view(B, S, self.num_labels)

    def forward(self, input_ids, segment_ids=None, input_mask=None, added_attention_mask=None, visual_embeds_mean=None,
                visual_embeds_att=None, temp=None, temp_lamb=None, lamb=None, labels=None, negative_rate=None, *,
                attention_mask=None, visual_feats=None, token_type_ids=None):
        """Positional form = the reference's (gate_cl_modeling.py:1319-1320).  Keyword aliases for the signature BASELINE.json's
        north_star names, ``forward(input_ids, attention_mask, visual_feats, ...)``: ``attention_mask`` = ``input_mask``,
        ``visual_feats`` = ``visual_embeds_att`` ([B,2048,7,7] or [B,R,2048]), ``token_type_ids`` = ``segment_ids``.  What such a
        caller leaves out gets the value the reference's feature builder gives it: segment ids all zero
        (My_cross_attention.py:362), every token valid, and ``added_attention_mask`` = ones[R] followed by the text mask (:373)."""
        if attention_mask is not None:
            if input_mask is not None:
                raise TypeError("forward() got both input_mask and its alias attention_mask")
            input_mask = attention_mask
        if visual_feats is not None:
            if visual_embeds_att is not None:
                raise TypeError("forward() got both visual_embeds_att and its alias visual_feats")
            visual_embeds_att = visual_feats
        if token_type_ids is not None:
            if segment_ids is not None:
                raise TypeError("forward() got both segment_ids and its alias token_type_ids")
            segment_ids = token_type_ids
        if visual_embeds_att is None:
            raise TypeError("forward() needs the region features (visual_embeds_att / visual_feats)")
        if segment_ids is None:
            segment_ids = torch.zeros_like(input_ids)
        if input_mask is None:
            input_mask = torch.ones_like(input_ids)
        if added_attention_mask is None:
            R = visual_embeds_att.shape[1] if visual_embeds_att.dim() == 3 else visual_embeds_att.shape[2] * visual_embeds_att.shape[3]
            added_attention_mask = torch.cat([torch.ones(input_ids.shape[0], R, dtype=input_mask.dtype, device=input_mask.device),
                                              input_mask], dim=1)
        logits = self.logits(input_ids, segment_ids, input_mask, added_attention_mask, visual_embeds_att)
        if labels is None:
            if self.crf is not None:   # cl_modeling.py:1386: pred_tags = self.crf.decode(feats, mask=input_mask.byte())
                return self.crf.decode(logits, mask=input_mask.byte())
            return logits
        if self.crf is not None:
            return -self.crf(logits, labels, mask=input_mask.byte(), reduction="mean")
        return token_ce_loss(logits, labels, input_mask, exact=_is_exact(self))


class MTCCMBertForMMTokenClassificationCRF_gate_1(BertPreTrainedModel):
    """the reference's simplest complete tagger
    (Cross_Modal_Interaction_Module.py:2383-2483): trunk -> ``x, _ = self.lstm(cross_output_layer)`` ->
    ``emissions = self.classifier(x)`` -> CRF loss (``reduction='token_mean'``) / Viterbi decode, selected by ``mode``.
    Same forward signature; the arguments the reference's forward never reads (input_ids, segment_ids, input_mask,
    clip_features, visual_embeds_mean, offsets, rela_score, ...) are accepted and ignored.  ``embedding`` may be an
    icka ``BertModel`` (default: built from ``config``); the absent RoBERTa ``last_encoder`` is not used by this
    class's forward and is kept only as an attribute."""

    def __init__(self, config, embedding=None, last_encoder=None, layer_num1=1, layer_num2=1, layer_num3=1,
                 num_labels=2):
        super().__init__(config)
        check_config(config)
        from .crf import CRF
        from .lstm import BiLSTM
        self.num_labels = num_labels
        self.last_encoder = last_encoder
        self.bert = embedding if embedding is not None else BertModel(config)
        self.hidden_size = config.hidden_size
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.vismap2text = nn.Linear(2048, config.hidden_size)
        self.txt2img_attention = BertCrossEncoder(config, layer_num1)
        self.lstm = BiLSTM(input_size=config.hidden_size, hidden_size=config.hidden_size, batch_first=True,
                           bidirectional=True)
        self.classifier = nn.Linear(config.hidden_size * 2, num_labels)
        self.crf = CRF(num_tags=num_labels, batch_first=True)
        if embedding is None:
            self.apply(self.init_bert_weights)
            self.lstm.reset_parameters()
            self.crf.reset_parameters()

    def emissions(self, ori_input_ids, ori_segment_ids, ori_input_mask, added_attention_mask, visual_embeds_att):
        B, S = ori_input_ids.shape
        A, _, _, cross, _ = _mner_trunk(self, ori_input_ids, ori_segment_ids, ori_input_mask, added_attention_mask,
                                        visual_embeds_att)
        x, _ = self.lstm(cross.view(B, S, self.hidden_size))
        if _is_exact(self):
            em = X.LinearFn.apply(A.anchor, x.reshape(B * S, 2 * self.hidden_size), self.classifier, A, False)
        else:
            em = ops.LinearFn.apply(A.anchor, x.reshape(B * S, 2 * self.hidden_size), self.classifier, A, True, K.EPI_NONE)
        return em.view(B, S, self.num_labels)

    def forward(self, input_ids, segment_ids, input_mask, ori_input_ids, ori_input_mask, ori_segment_ids,
                added_attention_mask, clip_features=None, visual_embeds_mean=None, visual_embeds_att=None, offsets=None,
                output_mask=None, rela_score=None, temp=None, temp_lamb=None, lamb=None, labels=None,
                negative_rate=None, mode=None):
        emissions = self.emissions(ori_input_ids, ori_segment_ids, ori_input_mask, added_attention_mask,
                                   visual_embeds_att)
        output_mask = (output_mask != 0)
        if mode == "train":
            return -self.crf(emissions, tags=labels, mask=output_mask, reduction="token_mean")
        if mode == "dev":
            pred_tags = self.crf.decode(emissions, mask=output_mask)
            return pred_tags, -self.crf(emissions, tags=labels, mask=output_mask, reduction="token_mean")
        if mode == "test":
            return self.crf.decode(emissions, mask=output_mask)
        return emissions


def token_ce_loss(logits: torch.Tensor, labels: torch.Tensor, input_mask: torch.Tensor, exact: bool = False) -> torch.Tensor:
    """Token-level cross-entropy, mean over tokens with input_mask != 0 (fused forward + logit gradient; ``exact``:
    f32 logit gradient instead of the bf16 one the bf16 classifier backward consumes)."""
    C = logits.shape[-1]
    lg = logits.reshape(-1, C)
    if lg.dtype != F32:
        raise TypeError("logits must be f32")
    if exact:
        return X.TokenCEFn.apply(lg.contiguous(), labels.reshape(-1).contiguous(), input_mask.reshape(-1).contiguous())
    return ops.TokenCEFn.apply(lg, labels.contiguous(), input_mask.contiguous())
