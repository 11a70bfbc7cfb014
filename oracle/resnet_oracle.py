"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch fp32, functional over a state_dict) of the reference's frozen
image encoder: resnet/resnet.py (Bottleneck.forward :74-93, ResNet._make_layer :121-136, ResNet.forward stem :139-142)
and resnet/resnet_utils.py (myResnet.forward :13-53).  Pinned against the reference's own classes by
tests/golden/make_golden_resnet.py (fixtures tests/golden/resnet_*.npz).  Only tests/ may import this module."""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


def _bn(x: Tensor, P: Params, prefix: str, eps: float = 1e-5) -> Tensor:
    """nn.BatchNorm2d in eval mode (running statistics)."""
    return F.batch_norm(x, P[prefix + ".running_mean"], P[prefix + ".running_var"], P[prefix + ".weight"],
                        P[prefix + ".bias"], False, 0.0, eps)


def bottleneck(P: Params, prefix: str, x: Tensor, stride: int) -> Tensor:
    """Bottleneck.forward (resnet/resnet.py:74-93)."""
    out = F.relu(_bn(F.conv2d(x, P[prefix + ".conv1.weight"]), P, prefix + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, P[prefix + ".conv2.weight"], stride=stride, padding=1), P, prefix + ".bn2"))
    out = _bn(F.conv2d(out, P[prefix + ".conv3.weight"]), P, prefix + ".bn3")
    residual = x
    if prefix + ".downsample.0.weight" in P:
        residual = _bn(F.conv2d(x, P[prefix + ".downsample.0.weight"], stride=stride), P, prefix + ".downsample.1")
    return F.relu(out + residual)


def resnet_features(P: Params, layers: Sequence[int], x: Tensor, prefix: str = "") -> Tensor:
    """conv1 -> bn1 -> relu -> maxpool -> layer1..4 (resnet/resnet.py:139-147; myResnet.forward :20-34)."""
    x = F.relu(_bn(F.conv2d(x, P[prefix + "conv1.weight"], stride=2, padding=3), P, prefix + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(layers):
        for bi in range(n):
            stride = 2 if (li > 0 and bi == 0) else 1
            x = bottleneck(P, "%slayer%d.%d" % (prefix, li + 1, bi), x, stride)
    return x


def my_resnet(P: Params, layers: Sequence[int], x: Tensor, att_size: int = 7, prefix: str = "") -> Tuple[Tensor, Tensor, Tensor]:
    """myResnet.forward (resnet/resnet_utils.py:13-53): (avgpool(7) flattened, fc = spatial mean, att)."""
    f = resnet_features(P, layers, x, prefix)
    fc = f.mean(3).mean(2)
    att = F.adaptive_avg_pool2d(f, [att_size, att_size])
    pooled = F.avg_pool2d(f, 7, stride=1).view(f.size(0), -1)
    return pooled, fc, att
