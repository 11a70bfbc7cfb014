"""Frozen ResNet image encoder on the HIP kernels, forward only.

Drop-in for the reference's ``resnet/resnet.py`` (``Bottleneck`` :57-93, ``ResNet`` :96-150, ``resnet152`` :204-213 --
same constructor arguments, attribute names and ``state_dict`` keys, so torchvision-style checkpoints load) and for
``resnet/resnet_utils.py::myResnet`` (:6-53: ``forward(x, att_size=7) -> (x, fc, att)``), which is what
My_cross_attention.py calls once per batch on the raw images with ``if_fine_tune=False``.

Every convolution is a GEMM of the GEMM kernels on NHWC bf16 activations: eval-mode BatchNorm is folded into the
(bf16) weights and an f32 bias when the module is first used (re-folded when a parameter changes), ReLU and the
residual add live in the GEMM epilogue, 3x3 / 7x7 convolutions go through patch matrices built by `icka_conv_*`.
The row count of every feature map is zero-padded to a multiple of 128 and channel counts to 64 (stem / layer1:
128x64 GEMM tiles) or multiples of 128, so that all GEMMs take the fast path.  There is no backward: the encoder is frozen in the reference
run (``fine_tune_cnn`` off) and ``if_fine_tune=True`` raises."""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import kernels as K

BF16, F32 = torch.bfloat16, torch.float32


def _pad128(n: int) -> int:
    return (n + 127) // 128 * 128


def _padc(c: int) -> int:
    """Channel padding: the 64-channel stem / layer1 tensors stay 64 wide (128x64 GEMM tiles), everything else is a
    multiple of 128."""
    return 64 if c <= 64 else _pad128(c)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise NotImplementedError("Bottleneck blocks run fused inside ResNet.features (GEMM + epilogue kernels)")


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        if block is not Bottleneck:
            raise ValueError("icka_amd.resnet implements the Bottleneck networks (resnet50/101/152)")
        self.inplanes = 64
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():   # resnet/resnet.py:113-119
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self._folded = None
        self._folded_key = None
        # 3x3 convolutions as implicit GEMMs (icka_conv3x3_gemm); False = patch matrix (icka_conv_im2col3x3) + GEMM
        self.implicit_conv = True
        # load_state_dict always invalidates the folded weights (version counters also move, but be explicit)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.refold())

    def refold(self) -> None:
        """Drop the folded conv+BatchNorm operands; the next forward re-folds.  The encoder is frozen in the reference
        (resnet/resnet_utils.py:13-53 under no_grad), so the fold is cached: it is re-done when a parameter / buffer
        version counter moved, after load_state_dict, after any optimizer step in the process, or after refold() --
        call refold() yourself after writing parameters through ``.data`` (invisible to version counters)."""
        self._folded = None
        self._folded_key = None

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------------------------------------ folding
    @staticmethod
    def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d, cin_pad: int, cout_pad: int, k_pad: int = 0):
        """conv + eval-mode BatchNorm -> (bf16 weight [cout_pad, taps*cin_pad (or k_pad)], f32 bias [cout_pad])."""
        w = conv.weight.detach().float()
        scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        bias = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
        w = w * scale[:, None, None, None]
        cout, cin, kh, kw = w.shape
        w = w.permute(0, 2, 3, 1)                                  # [cout, kh, kw, cin]: k = (ky*kw + kx)*cin + c
        if k_pad:                                                  # stem: taps*cin = 147 -> 192
            w2 = w.reshape(cout, kh * kw * cin)
            out = torch.zeros(cout_pad, k_pad, dtype=F32, device=w.device)
            out[:cout, :w2.shape[1]] = w2
        else:
            out = torch.zeros(cout_pad, kh, kw, cin_pad, dtype=F32, device=w.device)
            out[:cout, :, :, :cin] = w
            out = out.reshape(cout_pad, kh * kw * cin_pad)
        b = torch.zeros(cout_pad, dtype=F32, device=w.device)
        b[:cout] = bias
        return out.to(BF16).contiguous(), b.contiguous()

    def _prepare(self):
        from .arena import _OPT_STEPS
        key = (sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers()) + _OPT_STEPS[0],
               next(self.parameters()).device)
        if self.training:
            raise RuntimeError("icka_amd ResNet runs eval-mode BatchNorm only (call .eval(); the reference keeps the "
                               "encoder frozen)")
        if self._folded is not None and self._folded_key == key:
            return self._folded
        plan = {"stem": self._fold(self.conv1, self.bn1, 3, 64, k_pad=192), "blocks": []}
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                cin, p, cout = blk.conv1.in_channels, blk.conv1.out_channels, blk.conv3.out_channels
                entry = {"stride": blk.stride, "p": _padc(p), "cout": _padc(cout), "cin": _padc(cin),
                         "c1": self._fold(blk.conv1, blk.bn1, _padc(cin), _padc(p)),
                         "c2": self._fold(blk.conv2, blk.bn2, _padc(p), _padc(p)),
                         "c3": self._fold(blk.conv3, blk.bn3, _padc(p), _padc(cout)),
                         "down": None}
                if blk.downsample is not None:
                    entry["down"] = self._fold(blk.downsample[0], blk.downsample[1], _padc(cin), _padc(cout))
                plan["blocks"].append(entry)
        self._folded, self._folded_key = plan, key
        return plan

    def _zero_page(self, dev):
        z = getattr(self, "_zeros", None)
        if z is None or z.device != dev:
            z = self._zeros = torch.zeros(256, dtype=BF16, device=dev)
        return z

    # ------------------------------------------------------------------------------------------------ forward
    def features(self, x: torch.Tensor) -> Tuple[torch.Tensor, int, int, int]:
        """x f32 [B,3,H,W] on a ROCm device -> (last feature map as NHWC bf16 rows [rows_padded, 2048], B, Hf, Wf)."""
        if not x.is_cuda:
            raise TypeError("images must be on a ROCm device: icka_amd has no CPU path")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected images [B,3,H,W]")
        plan = self._prepare()
        x = x.float().contiguous()
        B, _, H, W = x.shape
        dev = x.device
        lib = K._lib.load()
        st = K._stream

        def gemm(a, wb, epi, aux=None):
            w, b = wb
            out = torch.empty(a.shape[0], w.shape[0], dtype=BF16, device=dev)
            K.gemm(K.GEMM_NT, a, w, out, bias=b, epilogue=epi, aux=aux)
            return out

        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        rows = _pad128(B * Ho * Wo)
        patches = torch.empty(rows, 192, dtype=BF16, device=dev)
        K.check(lib.icka_conv_stem_patches(x.data_ptr(), patches.data_ptr(), B, H, W, rows, st()), "icka_conv_stem_patches")
        cur = gemm(patches, plan["stem"], K.EPI_RELU)                       # conv1 + bn1 + relu (:139-141)
        Hc, Wc = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
        rows = _pad128(B * Hc * Wc)
        pooled = torch.empty(rows, 64, dtype=BF16, device=dev)
        K.check(lib.icka_conv_maxpool3x3s2(cur.data_ptr(), pooled.data_ptr(), B, Ho, Wo, 64, rows, st()),
                "icka_conv_maxpool3x3s2")                                    # maxpool (:142)
        cur = pooled
        for e in plan["blocks"]:                                             # Bottleneck.forward (:74-93)
            s = e["stride"]
            t1 = gemm(cur, e["c1"], K.EPI_RELU)
            Hn, Wn = (Hc + 2 - 3) // s + 1, (Wc + 2 - 3) // s + 1
            rows_out = _pad128(B * Hn * Wn)
            if self.implicit_conv:
                # conv2 + bn2 + relu as an implicit GEMM: the loader waves gather the 3x3 patches from t1 (no patch matrix)
                w2, b2 = e["c2"]
                t2 = torch.empty(rows_out, e["p"], dtype=BF16, device=dev)
                K.check(lib.icka_conv3x3_gemm(t1.data_ptr(), w2.data_ptr(), b2.data_ptr(), None, 0, t2.data_ptr(), B, Hc, Wc,
                                              e["p"], e["p"], s, rows_out, K.EPI_RELU, self._zero_page(dev).data_ptr(), st()),
                        "icka_conv3x3_gemm")
            else:
                pm = torch.empty(rows_out, 9 * e["p"], dtype=BF16, device=dev)
                K.check(lib.icka_conv_im2col3x3(t1.data_ptr(), pm.data_ptr(), B, Hc, Wc, e["p"], s, rows_out, st()),
                        "icka_conv_im2col3x3")
                t2 = gemm(pm, e["c2"], K.EPI_RELU)
            if e["down"] is not None:
                xs = cur
                if s != 1:
                    xs = torch.empty(rows_out, e["cin"], dtype=BF16, device=dev)
                    K.check(lib.icka_conv_subsample(cur.data_ptr(), xs.data_ptr(), B, Hc, Wc, e["cin"], s, rows_out, st()),
                            "icka_conv_subsample")
                res = gemm(xs, e["down"], K.EPI_NONE)
            else:
                res = cur
            cur = gemm(t2, e["c3"], K.EPI_ADD_RELU, aux=res)
            Hc, Wc = Hn, Wn
        return cur, B, Hc, Wc

    def forward(self, x):
        raise NotImplementedError("the ImageNet classification head is not used by ICKA; call features() or wrap the "
                                  "network in myResnet (resnet/resnet_utils.py)")


def resnet50(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kwargs)


def resnet101(pretrained=False, **kwargs):
    return ResNet(Bottleneck, [3, 4, 23, 3], **kwargs)


def resnet152(pretrained=False, **kwargs):
    """resnet/resnet.py:204-213 (``pretrained`` weights are loaded by the caller from a checkpoint, as the reference
    does with torch.load at My_cross_attention.py)."""
    return ResNet(Bottleneck, [3, 8, 36, 3], **kwargs)


class myResnet(nn.Module):
    """resnet/resnet_utils.py:6-53.  forward(x, att_size=7) -> (x [B,2048], fc [B,2048], att [B,2048,7,7]), all f32.
    ``last_tokens`` keeps the same features as bf16 region tokens [B*49, 2048] for the MNER trunk."""

    def __init__(self, resnet, if_fine_tune=False, device=None):
        super().__init__()
        if if_fine_tune:
            raise NotImplementedError("the image encoder is frozen (forward only); fine-tuning the CNN is out of scope")
        self.resnet = resnet
        self.if_fine_tune = if_fine_tune
        self.device = device
        self.last_tokens: Optional[torch.Tensor] = None

    @torch.no_grad()
    def forward(self, x, att_size=7):
        cur, B, Hf, Wf = self.resnet.features(x)
        if Hf != att_size or Wf != att_size:
            raise NotImplementedError("adaptive pooling to %dx%d from a %dx%d map (the reference feeds 224x224 images: "
                                      "7x7, where the pooling is the identity)" % (att_size, att_size, Hf, Wf))
        C, P = cur.shape[1], Hf * Wf
        att = torch.empty(B, C, Hf, Wf, dtype=F32, device=cur.device)
        fc = torch.empty(B, C, dtype=F32, device=cur.device)
        tokens = torch.empty(B * P, C, dtype=BF16, device=cur.device)
        K.check(K._lib.load().icka_conv_features_out(cur.data_ptr(), att.data_ptr(), fc.data_ptr(), tokens.data_ptr(),
                                                     B, P, C, K._stream()), "icka_conv_features_out")
        self.last_tokens = tokens
        return fc.clone(), fc, att      # x = avgpool(7) of a 7x7 map = the spatial mean = fc (:41-45)
