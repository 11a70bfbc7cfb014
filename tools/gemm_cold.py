#!/usr/bin/env python3
"""Diagnostic: one GEMM shape on COLD operands (a cycle of buffer sets larger than the 256 MB Infinity Cache) with the
operands' leading dimension padded by `pad` elements -- do row strides that are multiples of 4 KiB alias onto few
L2 / HBM channels?   usage: python tools/gemm_cold.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

BF16 = torch.bfloat16


def run(op, M, N, Kd, pad, nset, reps=6, share=None):
    """share: None, or "A" / "B" / "AB": that operand (or both, leaving only the output cold) is ONE buffer for all sets"""
    sets = []
    for _ in range(nset):
        if op == K.GEMM_NT:
            A = torch.randn(M, Kd + pad, device="cuda").to(BF16)[:, :Kd]
            B = torch.randn(N, Kd + pad, device="cuda").to(BF16)[:, :Kd]
        else:   # NN
            A = torch.randn(M, Kd + pad, device="cuda").to(BF16)[:, :Kd]
            B = torch.randn(Kd, N + pad, device="cuda").to(BF16)[:, :N]
        out = torch.empty(M, N + pad, dtype=BF16, device="cuda")[:, :N]
        if sets and share and "A" in share:
            A = sets[0][0]
        if sets and share and "B" in share:
            B = sets[0][1]
        sets.append((A, B, out))
    for A, B, o in sets:
        K.gemm(op, A, B, o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for A, B, o in sets:
            K.gemm(op, A, B, o)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nset)
    return us, 2.0 * M * N * Kd / us * 1e-6


for name, op, M, N, Kd in (("ffndn NT", K.GEMM_NT, 4096, 768, 3072), ("outproj NT", K.GEMM_NT, 4096, 768, 768),
                           ("ffnup NT", K.GEMM_NT, 4096, 3072, 768), ("qkv NT", K.GEMM_NT, 4096, 2304, 768),
                           ("dffnup NN", K.GEMM_NN, 4096, 768, 3072), ("dffndn NN", K.GEMM_NN, 4096, 3072, 768)):
    for nset in (1, 12):
        row = []
        for pad in (0, 64, 128):
            us, tf = run(op, M, N, Kd, pad, nset)
            row.append("pad %3d: %6.1f us %6.1f TF/s" % (pad, us, tf))
        print("%-10s %4dx%4dx%4d  %s  | %s" % (name, M, N, Kd, "warm (1 set) " if nset == 1 else "cold (12 sets)", "  ".join(row)), flush=True)
    us_a, _ = run(op, M, N, Kd, 0, 12, share="A")
    us_b, _ = run(op, M, N, Kd, 0, 12, share="B")
    print("%-10s which operand's coldness costs: A (activations) shared/warm, B cold: %.1f us | B (weights) shared/warm, A cold: %.1f us"
          % (name, us_a, us_b), flush=True)

# ---- epilogue cost on cold operands: the FFN pair as the step issues it (GELU with two outputs, GELU' with an aux read)
print("epilogue cost (12 cold sets, pad 0):")
for name, op, M, N, Kd, epi in (("ffnup NT", K.GEMM_NT, 4096, 3072, 768, K.EPI_GELU), ("dffndn NN", K.GEMM_NN, 4096, 3072, 768, K.EPI_DGELU),
                                ("dffnup NN", K.GEMM_NN, 4096, 768, 3072, K.EPI_ADD), ("ffndn NT", K.GEMM_NT, 4096, 768, 3072, K.EPI_NONE)):
    sets = []
    for _ in range(12):
        A = torch.randn(M, Kd, device="cuda").to(BF16)
        B = (torch.randn(N, Kd, device="cuda") if op == K.GEMM_NT else torch.randn(Kd, N, device="cuda")).to(BF16)
        out = torch.empty(M, N, dtype=BF16, device="cuda")
        outf = torch.empty(M, N, dtype=torch.float32, device="cuda")
        out2 = torch.empty(M, N, dtype=BF16, device="cuda")
        aux = torch.randn(M, N, device="cuda").to(BF16)
        bias = torch.randn(N, device="cuda")
        sets.append((A, B, out, out2, aux, bias, outf))

    def go(kind):
        for A, B, out, out2, aux, bias, outf in sets:
            if kind == "plain":
                K.gemm(op, A, B, out)
            elif kind == "plain_f32":
                K.gemm(op, A, B, outf)
            elif epi == K.EPI_GELU:
                K.gemm(op, A, B, out, bias=bias, epilogue=epi, out2=out2)
            elif epi in (K.EPI_DGELU, K.EPI_ADD):
                K.gemm(op, A, B, out, epilogue=epi, aux=aux)
            else:
                K.gemm(op, A, B, outf, bias=bias)
    res = []
    for kind in ("plain", "plain_f32", "epi"):
        go(kind)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            go(kind)
        e1.record()
        torch.cuda.synchronize()
        res.append("%s %.1f us" % (kind, e0.elapsed_time(e1) * 1e3 / 72))
    print("  %-10s %s" % (name, "   ".join(res)), flush=True)
