"""CPU oracle for the ICKA MNER hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain PyTorch (CPU, fp32, eager) restatement of the reference's
algorithm for the one hot path this repo accelerates: BERT text encoder ->
2048-d region projection -> text->image cross-attention -> gated fusion ->
per-token tag logits (+ token-level CE loss used by the benchmark harness).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / CPU baseline.  The product package
``icka_amd`` never imports anything from ``oracle/``.

Parity pinning: the reference (buctcurry/ICKA) ships no tests or golden vectors
for this path.  The oracle is therefore pinned against
outputs of the reference itself, imported in the dev container by
``tests/golden/make_golden.py`` (which also asserts oracle == reference to
<=1e-5) and committed as fixtures under ``tests/golden/``.

It is written functionally over a ``{state_dict key: tensor}`` mapping ``P`` so
that the same seeded weights drive the reference modules, this oracle and the
HIP product.  Each function cites the reference lines it follows
(paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


@dataclass
class OracleConfig:
    """The 10 fields the reference blocks read from their config object
    (Cross_Modal_Interaction_Module.py:45-60, a_transformers/configuration_bert.py:120-155)."""
    vocab_size: int = 30522
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    hidden_dropout_prob: float = 0.1
    attention_probs_dropout_prob: float = 0.1
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12


# --------------------------------------------------------------------------- primitives
def layer_norm(x: Tensor, P: Params, prefix: str, eps: float) -> Tensor:
    """TF-style LayerNorm, biased variance, eps inside the sqrt.
    Cross_Modal_Interaction_Module.py:518-522 (BertLayerNorm.forward)."""
    mu = x.mean(dim=-1, keepdim=True)
    var = (x - mu).pow(2).mean(dim=-1, keepdim=True)
    return P[prefix + ".weight"] * ((x - mu) / torch.sqrt(var + eps)) + P[prefix + ".bias"]


def gelu_erf(x: Tensor) -> Tensor:
    """Exact erf GELU, Cross_Modal_Interaction_Module.py:31-37."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def dense(x: Tensor, P: Params, prefix: str) -> Tensor:
    """nn.Linear with [out,in] weight."""
    return F.linear(x, P[prefix + ".weight"], P[prefix + ".bias"])


class MaskFeed(object):
    """Keep-multipliers for the dropout sites of ONE forward, in the order the sites run (nn.Dropout multiplies by
    0 or 1/(1-p): Cross_Modal_Interaction_Module.py:411, :500, :534, :563, :616, :953).  Passed in place of
    ``training=True``: every ``_drop`` call then multiplies by the next mask instead of drawing from the CPU RNG, so a
    train-mode step of the HIP path (whose masks come from its counter hash, exported with icka_dropout_mask /
    icka_attn_dropout_mask or restated in numpy) can be compared with the oracle end to end.  Site order of
    ``mner_logits``: embeddings; per BERT layer: attention probabilities, attention-output dense, FFN-output dense; the
    encoder output (:953); per cross layer: the same three.  ``masks`` is a list of tensors, or a callable
    ``(site_index, shape) -> tensor`` (masks made on demand: a whole bert-base step holds 1.3 GB of them)."""

    def __init__(self, masks):
        self.masks = masks
        self.used = 0

    def __bool__(self):
        return True

    def next(self, shape) -> Tensor:
        i = self.used
        self.used += 1
        m = self.masks(i, tuple(shape)) if callable(self.masks) else self.masks[i]
        if tuple(m.shape) != tuple(shape):
            raise ValueError("dropout site %d: mask shape %s, activation shape %s" % (i, tuple(m.shape), tuple(shape)))
        return m


def _drop(x: Tensor, p: float, training) -> Tensor:
    """nn.Dropout.  ``training`` is a bool, or a MaskFeed (train mode with the masks supplied by the caller)."""
    if isinstance(training, MaskFeed):
        return x * training.next(x.shape) if p > 0.0 else x
    return F.dropout(x, p, training) if (training and p > 0.0) else x


def additive_mask(mask01: Tensor, dtype=torch.float32) -> Tensor:
    """(1 - mask) * -10000.0 broadcast to [B,1,1,T].
    Cross_Modal_Interaction_Module.py:364-372 (text) and :962-965 (regions)."""
    ext = mask01[:, None, None, :].to(dtype)
    return (1.0 - ext) * -10000.0


def _split_heads(x: Tensor, heads: int) -> Tensor:
    b, t, hdim = x.shape
    return x.view(b, t, heads, hdim // heads).permute(0, 2, 1, 3)


def attention_core(P: Params, prefix: str, q_src: Tensor, kv_src: Tensor, add_mask: Tensor,
                   cfg: OracleConfig, training: bool) -> Tensor:
    """Q from q_src, K/V from kv_src; scores = QK^T, THEN / sqrt(d), THEN + mask; softmax; dropout; PV.
    BertSelfAttention.forward (Cross_Modal_Interaction_Module.py:478-506) when q_src is kv_src,
    BertCoAttention.forward (:590-624) otherwise."""
    h = cfg.num_attention_heads
    if cfg.hidden_size % h != 0:
        raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                         % (cfg.hidden_size, h))
    d = cfg.hidden_size // h
    q = _split_heads(dense(q_src, P, prefix + ".query"), h)
    k = _split_heads(dense(kv_src, P, prefix + ".key"), h)
    v = _split_heads(dense(kv_src, P, prefix + ".value"), h)
    scores = torch.matmul(q, k.transpose(-1, -2))
    scores = scores / math.sqrt(d)
    scores = scores + add_mask
    probs = torch.softmax(scores, dim=-1)
    probs = _drop(probs, cfg.attention_probs_dropout_prob, training)
    ctx = torch.matmul(probs, v)
    b, _, t, _ = ctx.shape
    return ctx.permute(0, 2, 1, 3).contiguous().view(b, t, cfg.hidden_size)


def dense_residual_norm(P: Params, prefix: str, x: Tensor, residual: Tensor, cfg: OracleConfig,
                        training: bool) -> Tensor:
    """LayerNorm(dropout(dense(x)) + residual): BertSelfOutput.forward (:561-565), BertOutput.forward (:532-536)."""
    y = _drop(dense(x, P, prefix + ".dense"), cfg.hidden_dropout_prob, training)
    return layer_norm(y + residual, P, prefix + ".LayerNorm", cfg.layer_norm_eps)


def feed_forward(P: Params, prefix: str, x: Tensor, cfg: OracleConfig, training: bool) -> Tensor:
    """BertIntermediate (:548-551) followed by BertOutput (:532-536)."""
    inter = gelu_erf(dense(x, P, prefix + ".intermediate.dense"))
    return dense_residual_norm(P, prefix + ".output", inter, x, cfg, training)


def bert_layer(P: Params, prefix: str, x: Tensor, add_mask: Tensor, cfg: OracleConfig, training: bool) -> Tensor:
    """BertLayer.forward (:438-442) = BertAttention (:451-454) + FFN."""
    ctx = attention_core(P, prefix + ".attention.self", x, x, add_mask, cfg, training)
    att = dense_residual_norm(P, prefix + ".attention.output", ctx, x, cfg, training)
    return feed_forward(P, prefix, att, cfg, training)


def cross_layer(P: Params, prefix: str, s1: Tensor, s2: Tensor, s2_add_mask: Tensor, cfg: OracleConfig,
                training: bool) -> Tensor:
    """BertCrossAttentionLayer.forward (:646-650): co-attention (Q=s1, K/V=s2), residual = s1, then FFN."""
    ctx = attention_core(P, prefix + ".attention.self", s1, s2, s2_add_mask, cfg, training)
    att = dense_residual_norm(P, prefix + ".attention.output", ctx, s1, cfg, training)
    return feed_forward(P, prefix, att, cfg, training)


def cross_encoder(P: Params, prefix: str, s1: Tensor, s2: Tensor, s2_add_mask: Tensor, cfg: OracleConfig,
                  layer_num: int, training: bool) -> List[Tensor]:
    """BertCrossEncoder.forward (:659-667); every layer re-uses the SAME s2."""
    outs = []
    for j in range(layer_num):
        s1 = cross_layer(P, "%s.layer.%d" % (prefix, j), s1, s2, s2_add_mask, cfg, training)
        outs.append(s1)
    return outs


def embeddings(P: Params, prefix: str, input_ids: Tensor, token_type_ids: Optional[Tensor], cfg: OracleConfig,
               training: bool) -> Tensor:
    """BertEmbeddings.forward (:398-412): word + position(arange) + type -> LayerNorm -> dropout.
    The word table is nn.Embedding(..., padding_idx=0) (:387): row 0 never receives gradient."""
    b, s = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    pos = torch.arange(s, dtype=torch.long, device=input_ids.device)
    e = (F.embedding(input_ids, P[prefix + ".word_embeddings.weight"], padding_idx=0)
         + P[prefix + ".position_embeddings.weight"][pos][None, :, :]
         + P[prefix + ".token_type_embeddings.weight"][token_type_ids])
    e = layer_norm(e, P, prefix + ".LayerNorm", cfg.layer_norm_eps)
    return _drop(e, cfg.hidden_dropout_prob, training)


def bert_model(P: Params, prefix: str, input_ids: Tensor, token_type_ids: Optional[Tensor],
               attention_mask: Optional[Tensor], cfg: OracleConfig, training: bool = False,
               all_layers: bool = False) -> Tuple[List[Tensor], Tensor]:
    """BertModel.forward (:353-382).  Returns (list of encoded layers, pooled)."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    add_mask = additive_mask(attention_mask)
    x = embeddings(P, prefix + ".embeddings", input_ids, token_type_ids, cfg, training)
    outs = []
    for i in range(cfg.num_hidden_layers):
        x = bert_layer(P, "%s.encoder.layer.%d" % (prefix, i), x, add_mask, cfg, training)
        if all_layers:
            outs.append(x)
    if not all_layers:
        outs.append(x)
    # BertPooler.forward (:675-681)
    pooled = torch.tanh(dense(x[:, 0], P, prefix + ".pooler.dense"))
    return outs, pooled


def region_tokens(visual_embeds_att: Tensor, regions: int) -> Tensor:
    """[B,2048,7,7] (myResnet 'att' output, resnet/resnet_utils.py:37-38,53) -> [B,R,2048] strided view:
    .view(-1, 2048, 49).permute(0, 2, 1)  (Cross_Modal_Interaction_Module.py:956, gate_cl_modeling.py:1328).
    A tensor already shaped [B,R,2048] (BASELINE synthetic layout) is passed through."""
    if visual_embeds_att.dim() == 3 and visual_embeds_att.shape[-1] == 2048:
        return visual_embeds_att
    return visual_embeds_att.reshape(-1, 2048, regions).permute(0, 2, 1)


# --------------------------------------------------------------------------- heads
def mner_trunk(P: Params, cfg: OracleConfig, input_ids: Tensor, segment_ids: Tensor, input_mask: Tensor,
               added_attention_mask: Tensor, visual_embeds_att: Tensor, layer_num1: int = 1, regions: int = 49,
               training: bool = False) -> Tuple[Tensor, Tensor, Tensor]:
    """Trunk shared by every MNER head: text encoder -> dropout -> region projection -> cross encoder.
    my_bert/cl_modeling.py:1341-1361, gate_cl_modeling.py:1322-1341, Cross_Modal_Interaction_Module.py:949-969.
    Returns (sequence_output, cross_output, pooled)."""
    outs, pooled = bert_model(P, "bert", input_ids, segment_ids, input_mask, cfg, training)
    seq = _drop(outs[-1], cfg.hidden_dropout_prob, training)
    vis = dense(region_tokens(visual_embeds_att, regions), P, "vismap2text")
    img_add = additive_mask(added_attention_mask[:, :regions])
    cross = cross_encoder(P, "txt2img_attention", seq, vis, img_add, cfg, layer_num1, training)[-1]
    return seq, cross, pooled


def gated_head_cl(P: Params, seq: Tensor, cross: Tensor) -> Tensor:
    """Gate = sigmoid(Gate_text(seq) + Gate_image(cross)); logits = classifier(cat(seq, Gate*cross)).
    my_bert/cl_modeling.py:1363-1371."""
    gate = torch.sigmoid(dense(seq, P, "Gate_text") + dense(cross, P, "Gate_image"))
    final = torch.cat((seq, gate * cross), dim=-1)
    return dense(final, P, "classifier")


def gated_head_gate_cl(P: Params, seq: Tensor, cross: Tensor) -> Tuple[Tensor, Tensor]:
    """gate_cl variant with the relevance score P (gate_cl_modeling.py:1364-1381; the two debug prints at
    :1340,:1342 and the .cuda() at :1345 are skipped).
    Returns (logits, crs_result)."""
    b = seq.shape[0]
    crs = dense(torch.cat((seq, cross), dim=-1).reshape(b, -1), P, "crs_classifier")
    rel = torch.softmax(crs, dim=-1)[:, -1][:, None, None]
    cross = rel * cross
    gate = torch.sigmoid(dense(seq, P, "Gate_text") + dense(cross, P, "Gate_image"))
    final = torch.cat((seq, gate * cross), dim=-1)
    return dense(final, P, "classifier"), crs


def scalar_gate_cross_modal(P: Params, cross: Tensor, token_embedding: Tensor, eps: float = 1e-5) -> Tensor:
    """Cross_Modal form: g = sigmoid(aux_head(cls_layer(cross[:,0], tok[:,0]))); g*tok + (1-g)*cross.
    Cross_Modal_Interaction_Module.py:879-884 (cls_layer_both: nn.LayerNorm default eps 1e-5) and :1029-1036.
    token_embedding comes from the out-of-scope RoBERTa stage and is an INPUT here."""
    feat = cross[:, 0] + token_embedding[:, 0]
    feat = F.layer_norm(feat, (feat.shape[-1],), P["cls_layer.proj_norm.weight"], P["cls_layer.proj_norm.bias"], eps)
    feat = dense(feat, P, "cls_layer.proj")
    g = torch.sigmoid(dense(feat, P, "aux_head")).view(-1, 1, 1)
    return g * token_embedding + (1.0 - g) * cross


def mner_logits(P: Params, cfg: OracleConfig, input_ids, segment_ids, input_mask, added_attention_mask,
                visual_embeds_att, layer_num1: int = 1, regions: int = 49, training: bool = False,
                variant: str = "cl") -> Tensor:
    """Per-token tag logits ('bert_feats'), my_bert/cl_modeling.py:1338-1371 ('cl') or
    gate_cl_modeling.py:1319-1381 ('gate_cl')."""
    seq, cross, _ = mner_trunk(P, cfg, input_ids, segment_ids, input_mask, added_attention_mask,
                               visual_embeds_att, layer_num1, regions, training)
    if variant == "cl":
        return gated_head_cl(P, seq, cross)
    if variant == "gate_cl":
        return gated_head_gate_cl(P, seq, cross)[0]
    raise ValueError(variant)


def token_ce_loss(logits: Tensor, labels: Tensor, input_mask: Tensor) -> Tensor:
    """Benchmark loss: token-level cross-entropy, mean over valid tokens.
    (The reference's CRF loss is outside the hot path.)"""
    c = logits.shape[-1]
    tgt = torch.where(input_mask.bool(), labels, torch.full_like(labels, -100))
    return F.cross_entropy(logits.reshape(-1, c).float(), tgt.reshape(-1), ignore_index=-100)


# --------------------------------------------------------------------------- key inventory
def hot_path_keys(cfg: OracleConfig, layer_num1: int = 1, num_labels: int = 13, with_crs: bool = False,
                  seq_len: int = 128) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys (and shapes) of the hot-path parameters, in the reference's naming."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    keys: Dict[str, Tuple[int, ...]] = {}

    def lin(prefix, o, i):
        keys[prefix + ".weight"] = (o, i)
        keys[prefix + ".bias"] = (o,)

    def ln(prefix):
        keys[prefix + ".weight"] = (H,)
        keys[prefix + ".bias"] = (H,)

    def block(prefix):
        for n in ("query", "key", "value"):
            lin("%s.attention.self.%s" % (prefix, n), H, H)
        lin(prefix + ".attention.output.dense", H, H)
        ln(prefix + ".attention.output.LayerNorm")
        lin(prefix + ".intermediate.dense", I, H)
        lin(prefix + ".output.dense", H, I)
        ln(prefix + ".output.LayerNorm")

    keys["bert.embeddings.word_embeddings.weight"] = (cfg.vocab_size, H)
    keys["bert.embeddings.position_embeddings.weight"] = (cfg.max_position_embeddings, H)
    keys["bert.embeddings.token_type_embeddings.weight"] = (cfg.type_vocab_size, H)
    ln("bert.embeddings.LayerNorm")
    for i in range(cfg.num_hidden_layers):
        block("bert.encoder.layer.%d" % i)
    lin("bert.pooler.dense", H, H)
    lin("vismap2text", H, 2048)
    for j in range(layer_num1):
        block("txt2img_attention.layer.%d" % j)
    lin("Gate_text", H, H)
    lin("Gate_image", H, H)
    lin("classifier", num_labels, 2 * H)
    if with_crs:
        lin("crs_classifier", 2, 2 * H * seq_len)
    return keys
