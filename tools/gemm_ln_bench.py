#!/usr/bin/env python3
"""Fused dense + LayerNorm launch (icka_gemm_ln) against icka_gemm (f32 out) + icka_ln_fwd on the c2 shapes, COLD operands
(12 buffer sets), variants interleaved round by round (median).  Third column: the fused launch with the statistics exchange
skipped (icka_gemm_ln_set_debug(1): wrong results, prices the exchange)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
lib = _lib.load()


def timed(fn, sets, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets:
            fn(s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))


for name, M, N, Kd in (("out-proj", 4096, 768, 768), ("ffn-down", 4096, 768, 3072)):
    sets = []
    for i in range(12):
        s = dict(A=torch.randn(M, Kd, device="cuda").to(BF16), W=(torch.randn(N, Kd, device="cuda") * Kd ** -0.5).to(BF16),
                 bias=torch.randn(N, device="cuda"), res=torch.randn(M, N, device="cuda"), gamma=torch.ones(N, device="cuda"),
                 beta=torch.zeros(N, device="cuda"), o=torch.empty(M, N, device="cuda"), y=torch.empty(M, N, dtype=BF16, device="cuda"),
                 yf=torch.empty(M, N, device="cuda"), xh=torch.empty(M, N, dtype=BF16, device="cuda"), rstd=torch.empty(M, device="cuda"))
        sets.append(s)

    def two(s):
        K.gemm(K.GEMM_NT, s["A"], s["W"], s["o"])
        K.ln_fwd(s["o"], s["bias"], s["res"], s["gamma"], s["beta"], s["y"], y_f32=s["yf"], xhat=s["xh"], rstd=s["rstd"], p_drop=0.1, seed=7)

    def gemm_only(s):
        K.gemm(K.GEMM_NT, s["A"], s["W"], s["o"])

    def fused(s):
        K.gemm_ln(s["A"], s["W"], s["bias"], s["res"], s["gamma"], s["beta"], s["y"], y_f32=s["yf"], xhat=s["xh"], rstd=s["rstd"], p_drop=0.1, seed=7)

    res = {k: [] for k in ("gemm", "gemm+ln", "fused", "fused-noexch", "fused-noepi", "fused-nores", "fused-nostore")}
    for rnd in range(7):
        for k, fn, dbg in (("gemm", gemm_only, 0), ("gemm+ln", two, 0), ("fused", fused, 0), ("fused-noexch", fused, 1),
                           ("fused-noepi", fused, 2), ("fused-nores", fused, 3), ("fused-nostore", fused, 4)):
            lib.icka_gemm_ln_set_debug(dbg)
            t = timed(fn, sets)
            if rnd:
                res[k].append(t)
    lib.icka_gemm_ln_set_debug(0)
    print("%-9s %dx%dx%d " % (name, M, N, Kd) + " | ".join("%s %.1f us" % (k, sorted(v)[len(v) // 2]) for k, v in res.items()), flush=True)
print("gemm_ln_error", K.gemm_ln_error())
