"""Blocks with the call convention of the reference's vendored HF-transformers BERT (a_transformers/modeling_bert.py),
for callers that import the block classes from that file (modeling/modeling_ensemble.py:8-11,
modeling/modeling_vcr_chunkalign_v10.py:9-13): keyword arguments, tuple returns, ``return_dict``.

The arithmetic and the ``state_dict`` keys are those of ``icka_amd.modeling`` (with identical
weights the two reference copies agree to 1e-6 and load each other's checkpoints with ``strict=True``); only the
encoder-only, absolute-position, no-cache branch of the reference exists here (a_transformers/modeling_bert.py:299-300,
:315, :333-353).  Everything the MNER path never exercises -- ``head_mask``, ``output_attentions``, decoder
cross-attention (``encoder_hidden_states``), key/value caches, relative position embeddings, feed-forward chunking --
raises ``NotImplementedError`` instead of being silently ignored.

    reference                                                     here
    BertEmbeddings.forward(input_ids, token_type_ids, ...) :184   BertEmbeddings
    BertSelfAttention.forward(hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
        encoder_attention_mask=None, past_key_value=None, output_attentions=False) -> tuple :267          BertSelfAttention
    BertSelfOutput / BertIntermediate / BertOutput.forward :362-452                                      same names
    BertAttention.forward(...) -> tuple :401                      BertAttention
    BertLayer.forward(...) -> tuple :468-534                      BertLayer
    BertEncoder.forward(..., return_dict=True) :543-631           BertEncoder
    BertPooler.forward :641                                       BertPooler
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import modeling as M
from .config import check_config

BertIntermediate = M.BertIntermediate
BertOutput = M.BertOutput
BertSelfOutput = M.BertSelfOutput
BertPooler = M.BertPooler


class BaseModelOutput(OrderedDict):
    """Minimal stand-in for transformers' ``BaseModelOutputWithPastAndCrossAttentions`` (a_transformers/
    modeling_outputs.py): attribute access, key access and integer indexing over the non-None fields."""

    def __init__(self, **fields):
        super().__init__((k, v) for k, v in fields.items() if v is not None)
        self._all = dict(fields)

    def __getattr__(self, name):
        try:
            return self.__dict__["_all"][name]
        except KeyError:
            raise AttributeError(name) from None

    def __getitem__(self, k):
        if isinstance(k, (int, slice)):
            return tuple(self.values())[k]
        return super().__getitem__(k)

    def to_tuple(self):
        return tuple(self.values())


def _unsupported(**kw) -> None:
    bad = [k for k, v in kw.items() if v]
    if bad:
        raise NotImplementedError(
            "%s: only the encoder-only / absolute-position / no-cache branch of a_transformers/modeling_bert.py is built "
            "(SURVEY.md section 2, row 7)" % ", ".join(bad))


def _check_hf_config(config) -> None:
    check_config(config)
    _unsupported(is_decoder=getattr(config, "is_decoder", False),
                 add_cross_attention=getattr(config, "add_cross_attention", False),
                 chunk_size_feed_forward=getattr(config, "chunk_size_feed_forward", 0),
                 relative_position_embeddings=getattr(config, "position_embedding_type", "absolute") != "absolute")


def _mask(attention_mask, B, S, device):
    """HF passes the extended additive mask [B,1,1,S] (get_extended_attention_mask) or None."""
    if attention_mask is None:
        return torch.zeros(B, 1, 1, S, dtype=torch.float32, device=device)
    return attention_mask


class BertEmbeddings(M.BertEmbeddings):
    """forward(input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0)."""

    def __init__(self, config):
        super().__init__(config)
        # exported when serialised, like the reference (:180-181): keeps strict state_dict loading both ways
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.position_embedding_type = getattr(config, "position_embedding_type", "absolute")
        if self.position_embedding_type != "absolute":
            raise NotImplementedError("relative position embeddings are not on the MNER path")

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None,
                past_key_values_length=0):
        _unsupported(position_ids=position_ids is not None, inputs_embeds=inputs_embeds is not None,
                     past_key_values_length=past_key_values_length)
        if input_ids is None:
            raise ValueError("input_ids is required")
        return super().forward(input_ids, token_type_ids)


class BertSelfAttention(M.BertSelfAttention):
    def __init__(self, config):
        _check_hf_config(config)
        super().__init__(config)

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False):
        _unsupported(head_mask=head_mask is not None, encoder_hidden_states=encoder_hidden_states is not None,
                     encoder_attention_mask=encoder_attention_mask is not None, past_key_value=past_key_value is not None,
                     output_attentions=output_attentions)
        B, S, _ = hidden_states.shape
        return (super().forward(hidden_states, _mask(attention_mask, B, S, hidden_states.device)),)


class BertAttention(M._IckaModule):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)
        self.pruned_heads = set()

    def prune_heads(self, heads):
        if heads:
            raise NotImplementedError("head pruning is not on the MNER path")

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False):
        self_outputs = self.self(hidden_states, attention_mask, head_mask, encoder_hidden_states, encoder_attention_mask,
                                 past_key_value, output_attentions)
        attention_output = self.output(self_outputs[0], hidden_states)
        return (attention_output,) + self_outputs[1:]


class BertLayer(M.BertLayer):
    """The fused layer kernels of ``icka_amd.modeling.BertLayer`` behind the HF signature; returns ``(layer_output,)``."""

    def __init__(self, config):
        _check_hf_config(config)
        super().__init__(config)
        self.chunk_size_feed_forward = 0
        self.seq_len_dim = 1
        self.is_decoder = False
        self.add_cross_attention = False

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_value=None, output_attentions=False):
        _unsupported(head_mask=head_mask is not None, encoder_hidden_states=encoder_hidden_states is not None,
                     encoder_attention_mask=encoder_attention_mask is not None, past_key_value=past_key_value is not None,
                     output_attentions=output_attentions)
        B, S, _ = hidden_states.shape
        return (super().forward(hidden_states, _mask(attention_mask, B, S, hidden_states.device)),)


class BertEncoder(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])

    def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                encoder_attention_mask=None, past_key_values=None, use_cache=None, output_attentions=False,
                output_hidden_states=False, return_dict=True):
        _unsupported(head_mask=head_mask is not None and any(h is not None for h in head_mask),
                     encoder_hidden_states=encoder_hidden_states is not None, past_key_values=past_key_values is not None,
                     use_cache=use_cache, output_attentions=output_attentions,
                     gradient_checkpointing=getattr(self.config, "gradient_checkpointing", False))
        all_hidden_states = () if output_hidden_states else None
        for layer_module in self.layer:
            if output_hidden_states:
                all_hidden_states = all_hidden_states + (hidden_states,)
            hidden_states = layer_module(hidden_states, attention_mask)[0]
        if output_hidden_states:
            all_hidden_states = all_hidden_states + (hidden_states,)
        if not return_dict:
            return tuple(v for v in (hidden_states, all_hidden_states) if v is not None)
        return BaseModelOutput(last_hidden_state=hidden_states, past_key_values=None, hidden_states=all_hidden_states,
                               attentions=None, cross_attentions=None)
