#!/usr/bin/env python3
"""One GEMM shape, cold operands, HIP events; knobs on the command line.   usage: python tools/gemm_one.py NT 8192 4096 1024 [sq=0|1|2]
(run under `rocprofv3 --kernel-trace --stats` to see which kernel serves the shape)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

op, M, N, Kd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = K._lib.load()
NSETS = 8
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    if k == "sets":
        NSETS = int(v)
        continue
    assert getattr(lib, {"sq": "icka_gemm_set_square_tiles", "wide": "icka_gemm_set_wide_tiles", "ring": "icka_gemm_set_ring", "w3p": "icka_gemm_set_persistent"}[k])(int(v)) == 0
BF16 = torch.bfloat16
sets = []
for _ in range(NSETS):
    A = torch.randn(M, Kd, device="cuda").to(BF16)
    B = (torch.randn(N, Kd, device="cuda") if op == "NT" else torch.randn(Kd, N, device="cuda")).to(BF16)
    sets.append((A, B, torch.empty(M, N, dtype=BF16, device="cuda")))
kop = K.GEMM_NT if op == "NT" else K.GEMM_NN
for A, B, o in sets:
    K.gemm(kop, A, B, o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    for A, B, o in sets:
        K.gemm(kop, A, B, o)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (10 * NSETS)
print("%s %dx%dx%d %s: %.1f us  %.1f TF/s" % (op, M, N, Kd, " ".join(sys.argv[5:]), us, 2.0 * M * N * Kd / us * 1e-6))
