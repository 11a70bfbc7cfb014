"""BertConfig: the configuration object the reference blocks are built from.

Mirrors the constructor and (de)serialisation helpers of the reference's ``BertConfig``
(Cross_Modal_Interaction_Module.py:45-139): first positional argument is either the vocabulary size or a path to
a JSON file; ``from_dict`` / ``from_json_file`` / ``to_dict`` / ``to_json_string`` behave the same way.
Any object exposing the same ten attributes is accepted by the modules (duck typing).
"""
from __future__ import annotations

import copy
import json


class BertConfig(object):
    def __init__(self, vocab_size_or_config_json_file, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1,
                 attention_probs_dropout_prob=0.1, max_position_embeddings=512, type_vocab_size=2,
                 initializer_range=0.02, layer_norm_eps=1e-12):
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                for key, value in json.loads(reader.read()).items():
                    self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size = hidden_size
            self.num_hidden_layers = num_hidden_layers
            self.num_attention_heads = num_attention_heads
            self.hidden_act = hidden_act
            self.intermediate_size = intermediate_size
            self.hidden_dropout_prob = hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings = max_position_embeddings
            self.type_vocab_size = type_vocab_size
            self.initializer_range = initializer_range
            self.layer_norm_eps = layer_norm_eps
        else:
            raise ValueError("First argument must be either a vocabulary size (int) "
                             "or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size_or_config_json_file=-1)
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def to_json_file(self, json_file_path):
        with open(json_file_path, "w", encoding="utf-8") as writer:
            writer.write(self.to_json_string())


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
