"""BertConfig: the configuration object the reference blocks are built from.

Mirrors the constructor and (de)serialisation helpers of the reference's ``BertConfig``
(Cross_Modal_Interaction_Module.py:45-139): first positional argument is either the vocabulary size or a path to
a JSON file; ``from_dict`` / ``from_json_file`` / ``to_dict`` / ``to_json_string`` behave the same way.
Any object exposing the same ten attributes is accepted by the modules (duck typing).
"""
from __future__ import annotations

import copy
import json


# field -> default, in the positional order of the reference constructor (after the vocabulary size / JSON path)
_FIELDS = (("hidden_size", 768), ("num_hidden_layers", 12), ("num_attention_heads", 12), ("intermediate_size", 3072),
           ("hidden_act", "gelu"), ("hidden_dropout_prob", 0.1), ("attention_probs_dropout_prob", 0.1),
           ("max_position_embeddings", 512), ("type_vocab_size", 2), ("initializer_range", 0.02), ("layer_norm_eps", 1e-12))


class BertConfig(object):
    """Plain attribute bag.  ``BertConfig(30522, hidden_size=768, ...)`` or ``BertConfig("config.json")``; fields arrive
    positionally or by keyword in the reference's order (``_FIELDS``); a JSON file may carry additional keys, which become
    attributes as well."""

    def __init__(self, vocab_size_or_config_json_file, *args, **kwargs):
        first = vocab_size_or_config_json_file
        if isinstance(first, str):
            self._absorb(_read_json(first))
            return
        if isinstance(first, bool) or not isinstance(first, int):
            raise ValueError("First argument must be either a vocabulary size (int) or the path to a pretrained model config "
                             "file (str)")
        if len(args) > len(_FIELDS):
            raise TypeError("BertConfig takes at most %d positional arguments" % (1 + len(_FIELDS)))
        values = dict(_FIELDS)
        values.update(zip((name for name, _ in _FIELDS), args))
        unknown = set(kwargs) - set(values)
        if unknown:
            raise TypeError("BertConfig got unexpected keyword arguments %s" % sorted(unknown))
        values.update(kwargs)
        self.vocab_size = first
        self._absorb(values)

    def _absorb(self, mapping) -> None:
        for key, value in mapping.items():
            setattr(self, key, value)

    @classmethod
    def from_dict(cls, json_object):
        config = cls(-1)
        config._absorb(json_object)
        return config

    @classmethod
    def from_json_file(cls, json_file):
        return cls.from_dict(_read_json(json_file))

    def to_dict(self):
        return copy.deepcopy(vars(self))

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def to_json_file(self, json_file_path):
        with open(json_file_path, "w", encoding="utf-8") as fh:
            fh.write(self.to_json_string())

    __repr__ = to_json_string


def _read_json(path):
    with open(path, "r", encoding="utf-8") as fh:
        return json.load(fh)


def check_config(config) -> None:
    """Validate the duck-typed fields the kernels depend on (raises ValueError like the reference, :459-462)."""
    if config.hidden_size % config.num_attention_heads != 0:
        raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                         % (config.hidden_size, config.num_attention_heads))
    act = getattr(config, "hidden_act", "gelu")
    if act != "gelu":
        raise ValueError("only the erf 'gelu' activation of the reference's BERT is implemented, got %r" % (act,))
    if config.hidden_size % 8 or config.intermediate_size % 8:
        raise ValueError("hidden_size and intermediate_size must be multiples of 8")


def check_head_size_bf16(config) -> None:
    """Kept for callers of the round-1 API.  The MFMA attention kernels are built for head size 64 (bert-base 768/12,
    bert-large 1024/16); since round 4 any other head size -- the reference accepts every hidden % heads == 0,
    Cross_Modal_Interaction_Module.py:459-462 -- runs the 16-bit path too, with the score / softmax / context core on the
    f32-input MFMA kernels of the fp32 mode (ops._attn_generic_fwd): nothing to refuse here any more."""
    if config.hidden_size // config.num_attention_heads <= 0:
        raise ValueError("bad head size")
