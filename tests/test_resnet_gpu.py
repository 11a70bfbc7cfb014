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


@pytest.mark.parametrize("B,H,W,C,Cout,stride", [(2, 56, 56, 64, 64, 1), (3, 28, 28, 128, 128, 2), (1, 14, 14, 256, 256, 1),
                                                 (2, 7, 7, 512, 512, 1), (1, 15, 9, 64, 128, 2), (5, 8, 8, 128, 64, 1)])
def test_implicit_conv3x3_equals_patch_matrix_gemm(B, H, W, C, Cout, stride):
    """icka_conv3x3_gemm (the loader waves gather the 3x3 patches) against icka_conv_im2col3x3 + icka_gemm and against
    torch's conv2d: borders, stride 2, odd image sizes, row padding, both tile widths, bias + ReLU / residual epilogues."""
    from icka_amd import kernels as K
    lib = K._lib.load()
    g = torch.Generator().manual_seed(B * 1000 + C)
    x = (torch.randn(B, H, W, C, generator=g) * 0.5).to(torch.bfloat16).cuda()          # NHWC
    w = (torch.randn(Cout, 3, 3, C, generator=g) * 0.05).to(torch.bfloat16).cuda()      # k = (ky*3+kx)*C + c
    bias = torch.randn(Cout, generator=g).cuda()
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    rows = B * Ho * Wo
    rp = (rows + 127) // 128 * 128
    zeros = torch.zeros(256, dtype=torch.bfloat16, device="cuda")
    aux = (torch.randn(rp, Cout, generator=g) * 0.5).to(torch.bfloat16).cuda()
    st = K._stream
    for epi, a in ((K.EPI_RELU, None), (K.EPI_ADD_RELU, aux), (K.EPI_NONE, None)):
        y = torch.full((rp, Cout), 7.0, dtype=torch.bfloat16, device="cuda")
        K.check(lib.icka_conv3x3_gemm(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None if a is None else a.data_ptr(),
                                      0 if a is None else a.stride(0), y.data_ptr(), B, H, W, C, Cout, stride, rp, epi,
                                      zeros.data_ptr(), st()), "icka_conv3x3_gemm")
        pm = torch.empty(rp, 9 * C, dtype=torch.bfloat16, device="cuda")
        K.check(lib.icka_conv_im2col3x3(x.data_ptr(), pm.data_ptr(), B, H, W, C, stride, rp, st()), "im2col")
        y2 = torch.empty(rp, Cout, dtype=torch.bfloat16, device="cuda")
        K.gemm(K.GEMM_NT, pm, w.view(Cout, 9 * C), y2, bias=bias, epilogue=epi, aux=a)
        assert torch.equal(y, y2), (epi, (y.float() - y2.float()).abs().max().item())
        ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, stride=stride,
                                         padding=1).permute(0, 2, 3, 1).reshape(rows, Cout)
        if a is not None:
            ref = ref + a[:rows].float()
        if epi != K.EPI_NONE:
            ref = ref.clamp_min(0)
        assert _rel(y[:rows].float(), ref) < 1e-2
