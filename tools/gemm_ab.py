#!/usr/bin/env python3
"""A/B of one library knob on the N = 768 GEMM shapes of the c2 step, COLD operands (12 buffer sets), variants
interleaved round by round in one process (median over rounds).  usage: gemm_ab.py knob v0 v1 [...]
knob: wide | ring | tile_n | direct"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16 = torch.bfloat16
lib = _lib.load()
knob = sys.argv[1]
vals = [int(v) for v in sys.argv[2:]]
field = {"wide": "wide_tiles", "ring": "ring", "tile_n": "tile_n", "direct": "direct_epilogue"}[knob]


def tune_of(v):      # the per-call tune word of a variant (0 = the library's own choice)
    return K.gemm_tune(**{field: v}) if (v or knob in ("wide", "direct")) else 0


def timed(fn, sets, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets:
            fn(*s)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * len(sets))


SHAPES = (("c4 outproj NT f32", "NT", 8192, 1024, 1024, True), ("c4 ffndn NT f32", "NT", 8192, 1024, 4096, True),
          ("c4 dffnup NN bf16", "NN", 8192, 1024, 4096, False), ("c4 dqkv NN bf16", "NN", 8192, 1024, 3072, False),
          ("c4 dout NN bf16", "NN", 8192, 1024, 1024, False), ("c5 ffndn NT f32", "NT", 8192, 768, 3072, True),
          ("c5 dffnup NN bf16", "NN", 8192, 768, 3072, False), ("c5 outproj NT f32", "NT", 8192, 768, 768, True),
          ("c4 ffnup NT bf16", "NT", 8192, 4096, 1024, False), ("c4 dffndn NN bf16", "NN", 8192, 4096, 1024, False),
          ("c5 ffnup NT bf16", "NT", 8192, 3072, 768, False), ("c5 qkv NT bf16", "NT", 8192, 2304, 768, False)) if os.environ.get("ICKA_AB_C4") else (
          ("outproj NT f32", "NT", 4096, 768, 768, True), ("ffndn NT f32", "NT", 4096, 768, 3072, True),
          ("dffnup NN bf16", "NN", 4096, 768, 3072, False), ("dqkv NN bf16", "NN", 4096, 768, 2304, False),
          ("dout NN bf16", "NN", 4096, 768, 768, False), ("ffnup NT bf16", "NT", 4096, 3072, 768, False))
for name, op, M, N, Kd, f32 in SHAPES:
    sets = []
    for _ in range(12):
        A = torch.randn(M, Kd, device="cuda").to(BF16)
        B = (torch.randn(N, Kd, device="cuda") if op == "NT" else torch.randn(Kd, N, device="cuda")).to(BF16)
        sets.append((A, B, torch.empty(M, N, dtype=torch.float32 if f32 else BF16, device="cuda")))
    kop = {"NT": K.GEMM_NT, "NN": K.GEMM_NN}[op]
    ref = None
    res = {v: [] for v in vals}
    for rnd in range(7):
        for v in vals:
            T = tune_of(v)
            t = timed(lambda A, B, o: K.gemm(kop, A, B, o, tune=T), sets)
            if rnd:
                res[v].append(t)
            out = sets[0][2].float().clone()
            if ref is None:
                ref = (sets[0][0].float() @ (sets[0][1].float().t() if op == "NT" else sets[0][1].float()))
            err = ((out - ref).norm() / ref.norm()).item()
            assert err < 1e-2, (name, v, err)
    fl = 2.0 * M * N * Kd
    line = "%-15s %4dx%4dx%4d " % (name, M, N, Kd)
    for v in vals:
        r = sorted(res[v])
        med = r[len(r) // 2]
        line += "| %s=%d %6.1f us (min %5.1f) %6.1f TF/s " % (knob, v, med, r[0], fl / med * 1e-6)
    print(line, flush=True)
