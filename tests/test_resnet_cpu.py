"""CPU: the ResNet oracle (oracle/resnet_oracle.py) against the fixtures produced by the reference's own resnet/ classes
(tests/golden/make_golden_resnet.py)."""
import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
from icka_amd import synth
from oracle import resnet_oracle as O


def _state(layers):
    from icka_amd.resnet import Bottleneck, ResNet
    net = ResNet(Bottleneck, list(layers))
    synth.fill_resnet_(net)
    return {k: v for k, v in net.state_dict().items()}


@pytest.mark.parametrize("name", ["resnet_tiny_1111_b2"])
def test_resnet_oracle_matches_reference_fixture(name):
    z = np.load(GOLDEN_DIR + "/" + name + ".npz")
    layers = [int(v) for v in z["layers"]]
    x = torch.randn(int(z["batch"]), 3, 224, 224, generator=torch.Generator().manual_seed(int(z["seed"])))
    with torch.no_grad():
        pooled, fc, att = O.my_resnet(_state(layers), layers, x)
    assert np.abs(fc.numpy() - z["fc"]).max() <= 1e-4 * np.abs(z["fc"]).max()
    assert np.abs(att[:, ::16].numpy() - z["att_sample"]).max() <= 1e-4 * float(z["att_abs_max"])
    assert torch.allclose(pooled, fc, rtol=1e-5, atol=1e-5)     # avgpool(7) of a 7x7 map is the spatial mean


def test_resnet_module_keeps_reference_state_dict_keys():
    from icka_amd.resnet import resnet152
    keys = set(resnet152().state_dict().keys())
    for k in ("conv1.weight", "bn1.running_var", "layer1.0.downsample.0.weight", "layer3.35.bn3.weight",
              "layer4.2.conv2.weight", "fc.bias"):
        assert k in keys
    assert len([k for k in keys if k.endswith("conv1.weight") or k.endswith("conv2.weight") or k.endswith("conv3.weight")
                or k.endswith("downsample.0.weight")]) == 155
