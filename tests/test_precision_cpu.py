"""CPU: which arithmetic mode a module runs in (icka_amd.resolved_precision).  VERDICT r02 #5: a user of bert-large must
not get the pure-bf16 mode -- which measures 2.2e-2 .. 2.5e-2 at 24 layers, above north_star's 2e-2 -- unless they ask for
it: the default ("auto") picks mixed16 for stacks deeper than 12 layers."""
import pytest

import icka_amd
from icka_amd.config import BertConfig
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF


def _model(layers, hidden=128):
    cfg = BertConfig(512, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=hidden // 64,
                     intermediate_size=2 * hidden, max_position_embeddings=64)
    return MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13)


def test_auto_precision_follows_the_depth_of_the_stack():
    base, deep = _model(12), _model(24)
    assert icka_amd.resolved_precision(base) == "bf16"
    assert icka_amd.resolved_precision(base.bert.encoder.layer[0]) == "bf16"
    assert icka_amd.resolved_precision(deep) == "mixed16"
    for m in (deep.bert.embeddings, deep.bert.encoder.layer[23], deep.txt2img_attention.layer[0]):
        assert icka_amd.resolved_precision(m) == "mixed16"


def test_explicit_precision_overrides_auto():
    deep = icka_amd.set_precision(_model(24), "bf16")
    assert icka_amd.resolved_precision(deep.bert.encoder.layer[0]) == "bf16"
    icka_amd.set_precision(deep, "fp32")
    assert icka_amd.resolved_precision(deep) == "fp32"
    icka_amd.set_precision(deep, "auto")
    assert icka_amd.resolved_precision(deep.bert.encoder.layer[5]) == "mixed16"
    with pytest.raises(ValueError):
        icka_amd.set_precision(deep, "fp64")
