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


def run(op, M, N, Kd, pad, nset, reps=6):
    sets = []
    for _ in range(nset):
        if op == K.GEMM_NT:
            A = torch.randn(M, Kd + pad, device="cuda").to(BF16)[:, :Kd]
            B = torch.randn(N, Kd + pad, device="cuda").to(BF16)[:, :Kd]
        else:   # NN
            A = torch.randn(M, Kd + pad, device="cuda").to(BF16)[:, :Kd]
            B = torch.randn(Kd, N + pad, device="cuda").to(BF16)[:, :N]
        out = torch.empty(M, N + pad, dtype=BF16, device="cuda")[:, :N]
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
