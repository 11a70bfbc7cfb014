"""GPU: the frozen ResNet image encoder (convolutions as GEMM kernels on NHWC bf16, folded BatchNorm) against the
fixtures made by the reference's resnet/ classes and against the CPU oracle run live."""
import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR
from icka_amd import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


@pytest.mark.parametrize("name", ["resnet_tiny_1111_b2", "resnet152_b1"])
def test_resnet_features_match_reference_fixture(name):
    from icka_amd.resnet import Bottleneck, ResNet, myResnet
    z = np.load(GOLDEN_DIR + "/" + name + ".npz")
    layers = [int(v) for v in z["layers"]]
    net = ResNet(Bottleneck, layers).eval()
    synth.fill_resnet_(net)
    enc = myResnet(net.cuda(), False, torch.device("cuda"))
    x = torch.randn(int(z["batch"]), 3, 224, 224, generator=torch.Generator().manual_seed(int(z["seed"])))
    pooled, fc, att = enc(x.cuda())
    assert att.dtype == torch.float32 and tuple(att.shape) == (int(z["batch"]), 2048, 7, 7)
    e_fc = _rel(fc.cpu(), torch.from_numpy(z["fc"]))
    e_att = _rel(att[:, ::16].cpu(), torch.from_numpy(z["att_sample"]))
    print("\n[%s] rel L2 err: fc %.3e, att %.3e" % (name, e_fc, e_att))
    assert e_fc < 2e-2 and e_att < 3e-2
    assert torch.equal(pooled, fc)
    tok = enc.last_tokens.float().view(int(z["batch"]), 49, 2048).permute(0, 2, 1).reshape(att.shape)
    assert torch.allclose(tok, att, rtol=0, atol=0)            # same bf16 values in the trunk's token layout


def test_resnet_batch_and_padding_against_live_oracle():
    """Batch 3 (row counts that are not multiples of 128 at every stage) against the oracle on the same weights."""
    from icka_amd.resnet import resnet50, myResnet
    from oracle import resnet_oracle as O
    net = resnet50().eval()
    synth.fill_resnet_(net)
    P = {k: v.clone() for k, v in net.state_dict().items()}
    x = torch.randn(3, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        _, rfc, ratt = O.my_resnet(P, [3, 4, 6, 3], x)
    enc = myResnet(net.cuda(), False, None)
    _, fc, att = enc(x.cuda())
    assert _rel(fc.cpu(), rfc) < 2e-2 and _rel(att.cpu(), ratt) < 3e-2
    # the encoder feeds the MNER trunk: [B,2048,7,7] att is what visual_embeds_att expects
    assert tuple(att.shape) == (3, 2048, 7, 7)


def test_resnet_refuses_unsupported_use():
    from icka_amd.resnet import resnet50, myResnet
    net = resnet50()
    with pytest.raises(NotImplementedError):
        myResnet(net, True, None)
    enc = myResnet(net.cuda().eval(), False, None)
    with pytest.raises(TypeError):
        enc(torch.zeros(1, 3, 224, 224))
    with pytest.raises(NotImplementedError):
        enc(torch.zeros(1, 3, 256, 256, device="cuda"))
    net.train()
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 3, 224, 224, device="cuda"))
