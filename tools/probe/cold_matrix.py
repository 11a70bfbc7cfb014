#!/usr/bin/env python3
"""Which operand's "coldness" costs the N = 768 GEMMs their +25-45 % inside the step?  ffn-down (NT 4096 x 768 x 3072, f32 out)
and d(ffn-up) (NN, bf16 out) timed with A, B, C each either the same buffer every launch or rotating over R buffers.
R = 2 keeps everything inside the 256 MB Infinity Cache, R = 12 does not."""
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icka_amd import kernels as K  # noqa: E402

BF16 = torch.bfloat16


def run(op, M, N, Kd, f32, rot, R, reps=6):
    nA, nB, nC = (R if r else 1 for r in rot)
    As = [torch.randn(M, Kd, device="cuda").to(BF16) for _ in range(nA)]
    Bs = [(torch.randn(N, Kd, device="cuda") if op == K.GEMM_NT else torch.randn(Kd, N, device="cuda")).to(BF16) for _ in range(nB)]
    Cs = [torch.empty(M, N, dtype=torch.float32 if f32 else BF16, device="cuda") for _ in range(nC)]
    n = 12 * reps
    for i in range(12):
        K.gemm(op, As[i % nA], Bs[i % nB], Cs[i % nC])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for i in range(n):
            K.gemm(op, As[i % nA], Bs[i % nB], Cs[i % nC])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n)
    return best


for name, op, M, N, Kd, f32 in (("ffn-down NT f32", K.GEMM_NT, 4096, 768, 3072, True), ("d(ffn-up) NN bf16", K.GEMM_NN, 4096, 768, 3072, False),
                                ("out-proj NT f32", K.GEMM_NT, 4096, 768, 768, True)):
    for R in (2, 12):
        line = "%-18s R=%2d |" % (name, R)
        for rot in itertools.product((0, 1), repeat=3):
            line += " A%dB%dC%d %5.1f" % (rot + (run(op, M, N, Kd, f32, rot, R),))
        print(line, flush=True)
