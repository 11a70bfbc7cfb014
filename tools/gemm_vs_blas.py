#!/usr/bin/env python3
"""Reference point: the c2 GEMM shapes on COLD operands (12 buffer sets > Infinity Cache), icka_gemm against torch.matmul
(hipBLASLt / rocBLAS behind it), plain bf16 outputs, HIP-event timed back to back.  Not used by the product."""
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


for name, op, M, N, Kd in (("qkv NT", "NT", 4096, 2304, 768), ("outproj NT", "NT", 4096, 768, 768), ("ffnup NT", "NT", 4096, 3072, 768),
                           ("ffndn NT", "NT", 4096, 768, 3072), ("dffndn NN", "NN", 4096, 3072, 768), ("dffnup NN", "NN", 4096, 768, 3072),
                           ("dqkv NN", "NN", 4096, 768, 2304), ("W1 TN", "TN", 3072, 768, 4096), ("Wqkv TN", "TN", 2304, 768, 4096)):
    sets = []
    for _ in range(12):
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
