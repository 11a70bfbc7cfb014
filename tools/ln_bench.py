#!/usr/bin/env python3
"""Diagnostic: the fused bias + dropout + residual + LayerNorm kernels at the c2 shape (4096 x 768), with and without the
dropout hash, forward and backward; cold inputs (12 buffer sets), 12 launches per captured graph, HIP events.
usage: python tools/ln_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
M, H, N = 4096, 768, 12
lib = _lib.load()
sets = []
for i in range(N):
    g = torch.Generator(device="cuda").manual_seed(i)
    sets.append(dict(x=torch.randn(M, H, device="cuda", generator=g), res=torch.randn(M, H, device="cuda", generator=g),
                     bias=torch.randn(H, device="cuda"), gamma=torch.ones(H, device="cuda"), beta=torch.zeros(H, device="cuda"),
                     y=torch.empty(M, H, dtype=BF16, device="cuda"), yf=torch.empty(M, H, dtype=F32, device="cuda"),
                     xhat=torch.empty(M, H, dtype=BF16, device="cuda"), rstd=torch.empty(M, dtype=F32, device="cuda"),
                     dy=torch.randn(M, H, device="cuda", generator=g).to(BF16), dres=torch.empty(M, H, dtype=BF16, device="cuda"),
                     dx=torch.empty(M, H, dtype=BF16, device="cuda"),
                     ws=torch.empty(lib.icka_ln_bwd_workspace_floats(H), dtype=F32, device="cuda")))


def fwd(s, p):
    K.ln_fwd(s["x"], s["bias"], s["res"], s["gamma"], s["beta"], s["y"], y_f32=s["yf"], xhat=s["xhat"], rstd=s["rstd"], p_drop=p,
             seed=1234)


def bwd(s, p):
    K.ln_bwd_slabs(s["dy"], s["xhat"], s["rstd"], s["gamma"], s["ws"], dres=s["dres"], dx=s["dx"], p_drop=p, seed=1234)


def timed(fn, p, reps=30):
    for s in sets:
        fn(s, p)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for s in sets:
            fn(s, p)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps / N


# rows a forward wave owns (> 1 overlaps next-row loads with stores): ICKA_TUNE_LN_ROWS_PER_WAVE, read once when the library loads
# -- one process per setting:  for r in 1 2 4 8; do ICKA_TUNE_LN_ROWS_PER_WAVE=$r python3 tools/ln_bench.py; done
print("ln_fwd, rows per wave %s: p = 0 %6.2f us | p = 0.1 %6.2f us"
      % (os.environ.get("ICKA_TUNE_LN_ROWS_PER_WAVE", "automatic"), timed(fwd, 0.0), timed(fwd, 0.1)), flush=True)
for name, fn in (("ln_fwd", fwd), ("ln_bwd (rows + column slabs)", bwd)):
    for rnd in range(2):
        a, b = timed(fn, 0.0), timed(fn, 0.1)
        print("%-30s round %d: p = 0 %6.2f us | p = 0.1 %6.2f us | dropout hash %+5.2f us per launch (launch boundary included)"
              % (name, rnd, a, b, b - a))
