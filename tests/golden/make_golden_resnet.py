#!/usr/bin/env python3
"""Generates tests/golden/resnet_*.npz from the REFERENCE's own image encoder (resnet/resnet.py + resnet/resnet_utils.py,
imported from /root/reference; dev container only) on by-key seeded weights (icka_amd.synth.seeded_resnet_tensor), and
asserts that oracle/resnet_oracle.py reproduces it.  The fixtures hold inputs' seeds and expected outputs only.
usage: python tests/golden/make_golden_resnet.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from icka_amd import synth  # noqa: E402
from oracle import resnet_oracle as O  # noqa: E402
from resnet import resnet as R  # noqa: E402  (reference)
from resnet.resnet_utils import myResnet  # noqa: E402  (reference)


def images(B, H, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 3, H, H, generator=g)


def case(name, layers, B, seed):
    net = R.ResNet(R.Bottleneck, layers).eval()
    synth.fill_resnet_(net)
    enc = myResnet(net, False, torch.device("cpu"))
    x = images(B, 224, seed)
    with torch.no_grad():
        pooled, fc, att = enc(x)
        P = {k: v for k, v in net.state_dict().items()}
        op, ofc, oatt = O.my_resnet(P, layers, x)
    assert (oatt - att).abs().max().item() <= 1e-5 * att.abs().max().item(), "oracle != reference (att)"
    assert (ofc - fc).abs().max().item() <= 1e-5 * fc.abs().max().item(), "oracle != reference (fc)"
    assert (op - pooled).abs().max().item() <= 1e-5 * pooled.abs().max().item()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), layers=np.array(layers), batch=B,
                        seed=seed, fc=fc.numpy(), att_sample=att[:, ::16].numpy(), att_abs_max=att.abs().max().item(),
                        att_l2=att.norm().item())
    print(name, "att max", att.abs().max().item(), "mean", att.mean().item())


if __name__ == "__main__":
    case("resnet_tiny_1111_b2", [1, 1, 1, 1], 2, 11)
    case("resnet152_b1", [3, 8, 36, 3], 1, 12)
