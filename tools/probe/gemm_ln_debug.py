#!/usr/bin/env python3
"""Debug aid: where does icka_gemm_ln differ from icka_gemm + icka_ln_fwd?  (python3 tools/probe/gemm_ln_debug.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icka_amd import kernels as k  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
M, N, K, p, seed = 4096, 768, 768, 0.1, 0x1234567890
g = torch.Generator(device="cuda").manual_seed(1)
h = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(BF16)
w = (torch.randn(N, K, device="cuda", generator=g) * 0.5).to(BF16)
bias, gamma, beta = torch.randn(N, device="cuda"), torch.randn(N, device="cuda") + 1, torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda")


def outs():
    return (torch.full((M, N), float("nan"), dtype=F32, device="cuda"), torch.empty(M, N, dtype=BF16, device="cuda"),
            torch.empty(M, N, dtype=F32, device="cuda"), torch.empty(M, N, dtype=BF16, device="cuda"), torch.empty(M, dtype=F32, device="cuda"))


o0, y0, t0, xh0, rs0 = outs()
k.gemm(k.GEMM_NT, h, w, o0)
k.ln_fwd(o0, bias, res, gamma, beta, y0, xhat=xh0, rstd=rs0, p_drop=p, seed=seed, y_f32=t0)
sync = k.gemm_ln_sync("cuda")
for rep in range(3):
    o1, y1, t1, xh1, rs1 = outs()
    ok = k.gemm_ln(h, w, o1, bias, res, gamma, beta, y1, sync, xhat=xh1, rstd=rs1, p_drop=p, seed=seed, y_f32=t1)
    torch.cuda.synchronize()
    for name, a, b in (("o", o1, o0), ("y", y1, y0), ("twin", t1, t0), ("xhat", xh1, xh0), ("rstd", rs1.view(-1, 1), rs0.view(-1, 1))):
        a, b = a.float(), b.float()
        bad = ~((a == b) | (a.isnan() & b.isnan()))
        rows = bad.any(1).nonzero().view(-1)
        cols = bad.any(0).nonzero().view(-1)
        print("rep %d %-5s launched %s  mismatches %8d  nan %8d  rows %s..%s (%d)  cols %s..%s (%d)  max|d| %.3e" % (
            rep, name, ok, int(bad.sum()), int(a.isnan().sum()), rows[:1].tolist(), rows[-1:].tolist(), rows.numel(),
            cols[:1].tolist(), cols[-1:].tolist(), cols.numel(), float((a - b).nan_to_num().abs().max())))
    k.gemm_ln_check_error("debug")
    print("   counter residue", int(sync.abs().sum()))
