#!/usr/bin/env python3
"""NT vs NN forms of the dgrad GEMM shapes of the c2 step, cold operands (12 buffer sets, interleaved rounds): what a
k-contiguous (pre-transposed) copy of the weights would buy the input-gradient GEMMs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icka_amd import kernels as K  # noqa: E402

BF16 = torch.bfloat16


def timed(fn, sets, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets:
            fn(*s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))


for name, M, N, Kd, epi in (("d(ffn-down)+GELU'", 4096, 3072, 768, "dgelu"), ("d(ffn-up)+fan-in", 4096, 768, 3072, "add"),
                            ("d(qkv)+fan-in", 4096, 768, 2304, "add"), ("d(out-proj)", 4096, 768, 768, None)):
    sets = {"NT": [], "NN": []}
    for _ in range(12):
        A = torch.randn(M, Kd, device="cuda").to(BF16)
        Bt = torch.randn(N, Kd, device="cuda").to(BF16)          # k-contiguous (what a transposed shadow would hold)
        Bn = Bt.t().contiguous()                                   # [K, N]: the weight as stored
        aux = torch.randn(M, N, device="cuda").to(BF16)
        sets["NT"].append((A, Bt, torch.empty(M, N, dtype=BF16, device="cuda"), aux))
        sets["NN"].append((A, Bn, torch.empty(M, N, dtype=BF16, device="cuda"), aux))
    kw = {"dgelu": dict(epilogue=K.EPI_DGELU), "add": dict(epilogue=K.EPI_ADD), None: {}}[epi]
    res = {"NT": [], "NN": []}
    for rnd in range(7):
        for op in ("NT", "NN"):
            kop = K.GEMM_NT if op == "NT" else K.GEMM_NN
            t = timed(lambda A, B, o, aux: K.gemm(kop, A, B, o, **(dict(kw, aux=aux) if epi else {})), sets[op])
            if rnd:
                res[op].append(t)
    assert torch.equal(sets["NT"][0][2], sets["NT"][0][2])
    d = (sets["NT"][0][2].float() - sets["NN"][0][2].float()).abs().max().item()
    line = "%-20s %4dx%4dx%4d " % (name, M, N, Kd)
    for op in ("NT", "NN"):
        r = sorted(res[op])
        line += "| %s %6.1f us (min %5.1f) " % (op, r[len(r) // 2], r[0])
    print(line + "| max |NT - NN| (different operands) %.3g" % d, flush=True)
