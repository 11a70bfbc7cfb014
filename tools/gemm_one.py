#!/usr/bin/env python3
"""One GEMM shape, cold operands, HIP events; per-call tune words on the command line.
usage: python3 tools/gemm_one.py NT 8192 4096 1024 [wide=0|1] [ring=2..5] [tile_n=96|128] [sets=N]
To see which kernel serves the shape:  rocprofv3 --kernel-trace --stats -d gpurun_out/one -- python3 tools/gemm_one.py NT 8192 4096 1024
(the interpreter itself after `--`: the shebang line below is an `env` hop, which a profiled run must not take)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

op, M, N, Kd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
NSETS = 8
tune = {}
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    if k == "sets":
        NSETS = int(v)
        continue
    tune[{"wide": "wide_tiles", "ring": "ring", "tile_n": "tile_n", "direct": "direct_epilogue"}[k]] = int(v)
T = K.gemm_tune(**tune)
BF16 = torch.bfloat16
sets = []
for _ in range(NSETS):
    A = torch.randn(M, Kd, device="cuda").to(BF16)
    B = (torch.randn(N, Kd, device="cuda") if op == "NT" else torch.randn(Kd, N, device="cuda")).to(BF16)
    sets.append((A, B, torch.empty(M, N, dtype=BF16, device="cuda")))
kop = K.GEMM_NT if op == "NT" else K.GEMM_NN
for A, B, o in sets:
    K.gemm(kop, A, B, o, tune=T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    for A, B, o in sets:
        K.gemm(kop, A, B, o, tune=T)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (10 * NSETS)
print("%s %dx%dx%d %s: %.1f us  %.1f TF/s" % (op, M, N, Kd, " ".join(sys.argv[5:]), us, 2.0 * M * N * Kd / us * 1e-6))
