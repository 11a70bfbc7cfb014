"""CPU oracle for the reference's current (published) ICKA tagger  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Plain PyTorch (CPU, fp32, eager) restatement of Cross_Modal_Interaction_Module.MTCCMBertForMMTokenClassificationCRF
.forward (/root/reference/Cross_Modal_Interaction_Module.py:941-1057), functional over a ``{state_dict key: tensor}``
mapping like oracle/mner_oracle.py.  Only tests/ may import it.

Pinning.  Everything except ``last_encoder`` is pinned against the reference's own forward: tests/golden/
make_golden_cross_modal.py imports the reference class, injects this file's ``PromptRobertaOracle`` as its
``last_encoder`` argument (the reference takes it as a constructor argument, :888) and a thin wrapper of the reference's
own BertModel as ``embedding``, runs the reference forward to the emissions its CRF receives, asserts this oracle equals
them to 1e-5 and commits the fixture.  ``last_encoder`` itself is a third-party dependency that is NOT in the reference
tree (``local_transformers.adapter_transformers.models.roberta_ner``, un-pinned, imported at My_cross_attention.py:4):
PARITY UNPINNED for that stage.  ``prompt_roberta`` below states the call contract the reference does fix (argument
names :1010-1012, output length :1014, window arithmetic :1022-1024, prompt order :1001, mask-token positions from the
token dump at My_cross_attention.py:402-404) on a standard RoBERTa encoder (HF RobertaEmbeddings conventions:
padding_idx 1, positions from padding_idx+1, one token type, LayerNorm eps 1e-5; RoBERTa's layer is the BERT layer).
The BiLSTM is ATen's nn.LSTM, the CRF is oracle/crf_oracle.py (pytorch-crf 0.7.2 restated; also unpinned).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import mner_oracle as O

Tensor = torch.Tensor
Params = O.Params


def prompt_mapping(P: Params, prefix: str, x: Tensor) -> Tensor:
    """mapping_network_alignment / mapping_network_vision in eval mode (:914-928):
    Sequential(Dropout, Linear[1], Tanh, Dropout, Linear[4])."""
    return O.dense(torch.tanh(O.dense(x, P, prefix + ".1")), P, prefix + ".4")


def splice_index(s_in: int, n_prompt: int, mask_positions: Sequence[int] = (3, 11)) -> List[int]:
    """>= 0: token position; < 0: prompt vector -1-v.  The prompt vectors are dealt to the mask positions in equal
    consecutive shares (vision prefix first, :1001)."""
    share, out, j = n_prompt // len(mask_positions), [], 0
    for t in range(s_in):
        if t in mask_positions:
            out.extend(-1 - (j + i) for i in range(share))
            j += share
        else:
            out.append(t)
    return out


def prompt_roberta(P: Params, prefix: str, cfg: O.OracleConfig, input_ids: Tensor, attention_mask: Tensor,
                   prompt_embeddings: Tensor, prompt_mask: Tensor, mask_positions: Sequence[int] = (3, 11),
                   padding_idx: int = 1) -> Tensor:
    """The prompt-accepting encoder stage (module docstring).  -> [B, S_in - n_mask + P, H]."""
    b, s_in = input_ids.shape
    idx = torch.tensor(splice_index(s_in, prompt_embeddings.shape[1], mask_positions), dtype=torch.long)
    tok = idx >= 0
    words = F.embedding(input_ids, P[prefix + ".embeddings.word_embeddings.weight"], padding_idx=padding_idx)
    x = torch.where(tok[None, :, None], words[:, idx.clamp(min=0)], prompt_embeddings[:, (-1 - idx).clamp(min=0)])
    s = idx.shape[0]
    pos = torch.arange(s, dtype=torch.long) + padding_idx + 1
    x = x + P[prefix + ".embeddings.position_embeddings.weight"][pos][None] \
        + P[prefix + ".embeddings.token_type_embeddings.weight"][0][None, None]
    x = O.layer_norm(x, P, prefix + ".embeddings.LayerNorm", cfg.layer_norm_eps)
    mask = torch.where(tok[None, :], attention_mask[:, idx.clamp(min=0)], prompt_mask[:, (-1 - idx).clamp(min=0)])
    add_mask = O.additive_mask(mask)
    for i in range(cfg.num_hidden_layers):
        x = O.bert_layer(P, "%s.encoder.layer.%d" % (prefix, i), x, add_mask, cfg, False)
    return x


class PromptRobertaOracle(torch.nn.Module):
    """``prompt_roberta`` behind the call signature the reference uses (:1010-1012), so that the reference's own
    forward can run with it as ``last_encoder`` (golden generation).  Holds NO parameters of its own: reads ``P``."""

    def __init__(self, P: Params, prefix: str, cfg: O.OracleConfig, mask_positions: Sequence[int] = (3, 11)):
        super().__init__()
        self.P, self.prefix, self.cfg, self.mask_positions = P, prefix, cfg, tuple(mask_positions)

    def forward(self, input_ids=None, token_type_ids=None, attention_mask=None, prompt_embeddings=None,
                input_mask=None, offset=None):
        return (prompt_roberta(self.P, self.prefix, self.cfg, input_ids, attention_mask, prompt_embeddings,
                               input_mask, self.mask_positions),)


def bilstm(P: Params, prefix: str, x: Tensor) -> Tensor:
    """nn.LSTM(H, H, batch_first=True, bidirectional=True) (:905-908) -- ATen, not reference code."""
    h = x.shape[-1]
    lstm = torch.nn.LSTM(h, h, batch_first=True, bidirectional=True)
    sd = {k[len(prefix) + 1:]: v for k, v in P.items() if k.startswith(prefix + ".")}
    return torch.func.functional_call(lstm, sd, (x,))[0]


def emissions(P: Params, cfg: O.OracleConfig, cfg_r: O.OracleConfig, input_ids: Tensor, input_mask: Tensor,
              ori_input_ids: Tensor, ori_input_mask: Tensor, ori_segment_ids: Tensor, added_attention_mask: Tensor,
              clip_features: Tensor, visual_embeds_mean: Tensor, visual_embeds_att: Tensor, offset: int,
              layer_num1: int = 1, prompt_len: int = 5, window: int = 128,
              mask_positions: Sequence[int] = (3, 11)) -> Tuple[Tensor, dict]:
    """Forward of the reference model up to the emissions the CRF receives (:949-1043), eval mode.
    Returns (emissions [B,S,C], intermediates)."""
    b = ori_input_ids.shape[0]
    # :950-969 trunk (text encoder -> dropout -> vismap2text -> txt2img cross encoder), regions = 49
    seq, cross, _ = O.mner_trunk(P, cfg, ori_input_ids, ori_segment_ids, ori_input_mask, added_attention_mask,
                                 visual_embeds_att, layer_num1, 49, False)
    # :954 CLIP token -> hidden;  :976-989 two single-query cross encoders over the text tokens
    clip = O.dense(clip_features.float().squeeze(1), P, "vismapping").unsqueeze(1)
    text_add = O.additive_mask(ori_input_mask)
    for i in range(2):
        clip = O.cross_encoder(P, "cls_layer_Y.%d" % i, clip, cross, text_add, cfg, layer_num1, False)[-1]
    # :995-1004 prompts
    align = prompt_mapping(P, "mapping_network_alignment", clip).unsqueeze(1).view(b, prompt_len, -1)
    vision = prompt_mapping(P, "mapping_network_vision", visual_embeds_mean.float()).reshape(b, prompt_len, -1)
    prefix = torch.cat([vision, align], dim=1)
    if prefix.shape[2] != 1024:
        prefix = O.dense(prefix, P, "lastproj")
    # :1006-1013 prompt-accepting encoder
    pm = input_mask[:, :1].repeat(1, 2 * prompt_len)
    enc = prompt_roberta(P, "last_encoder", cfg_r, input_ids, input_mask, prefix, pm, mask_positions)
    off2 = offset - 2 + prefix.shape[1]                                    # :1022
    tok = enc[:, off2:off2 + window, :]                                    # :1024
    # :1029-1036 scalar gate + blend;  :1042-1043 BiLSTM + classifier
    result = O.scalar_gate_cross_modal(P, cross, tok)
    x = bilstm(P, "lstm", result)
    em = O.dense(x, P, "classifier")
    return em, {"cross": cross, "clip": clip, "prefix": prefix, "enc": enc, "result": result}
