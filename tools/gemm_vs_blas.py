#!/usr/bin/env python3
"""Reference point: the GEMM shapes of a BASELINE configuration on COLD operands (buffer sets > Infinity Cache), icka_gemm
against torch.matmul (hipBLASLt / rocBLAS behind it), plain bf16 outputs (f32 for the weight gradients), HIP-event timed back
to back.  Not used by the product.   usage: python tools/gemm_vs_blas.py [c2|c4|c5]   (c4: bert-large, M = 32 x 256 = 8192;
c5: bert-base at batch 64, M = 8192)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

BF16 = torch.bfloat16


def timed(fn, sets, reps=6):
    for s in sets:
        fn(*s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets:
            fn(*s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))


cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
M, H = {"c2": (4096, 768), "c4": (8192, 1024), "c5": (8192, 768)}[cfg]
I = 4 * H
NSETS = 12 if M * I * 2 * 12 < (3 << 30) else 8
print("# %s: M = %d tokens, H = %d, I = %d; %d rotating buffer sets per shape" % (cfg, M, H, I, NSETS))
for name, op, M, N, Kd in (("qkv NT", "NT", M, 3 * H, H), ("outproj NT", "NT", M, H, H), ("ffnup NT", "NT", M, I, H),
                           ("ffndn NT", "NT", M, H, I), ("dffndn NN", "NN", M, I, H), ("dffnup NN", "NN", M, H, I),
                           ("doutproj NN", "NN", M, H, H), ("dqkv NN", "NN", M, H, 3 * H), ("W1 TN", "TN", I, H, M),
                           ("W2 TN", "TN", H, I, M), ("Wqkv TN", "TN", 3 * H, H, M), ("Wo TN", "TN", H, H, M)):
    sets = []
    for _ in range(NSETS):
        if op == "NT":
            A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(N, Kd, device="cuda").to(BF16)
        elif op == "NN":
            A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
        else:
            A, B = torch.randn(Kd, M, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
        sets.append((A, B, torch.empty(M, N, dtype=BF16 if op != "TN" else torch.float32, device="cuda")))
    kop = {"NT": K.GEMM_NT, "NN": K.GEMM_NN, "TN": K.GEMM_TN}[op]
    mine = timed(lambda A, B, o: K.gemm(kop, A, B, o), sets)
    if op == "NT":
        blas = timed(lambda A, B, o: torch.matmul(A, B.t(), out=o), sets)
    elif op == "NN":
        blas = timed(lambda A, B, o: torch.matmul(A, B, out=o), sets)
    else:
        sets2 = [(A, B, torch.empty(M, N, dtype=BF16, device="cuda")) for A, B, _ in sets]
        blas = timed(lambda A, B, o: torch.matmul(A.t(), B, out=o), sets2)
    fl = 2.0 * M * N * Kd
    print("%-11s %4dx%4dx%4d  icka %6.1f us %6.1f TF/s | torch.matmul %6.1f us %6.1f TF/s"
          % (name, M, N, Kd, mine, fl / mine * 1e-6, blas, fl / blas * 1e-6), flush=True)
