#!/usr/bin/env python3
"""Diagnostic: per-tile error map of the fast-path GEMM for a forced tile width (the ICKA_TUNE_TILE_N word of the call's descriptor)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
lib = _lib.load()
bn = int(sys.argv[1]) if len(sys.argv) > 1 else 96
T = K.gemm_tune(tile_n=bn, **({"ring": int(sys.argv[2])} if len(sys.argv) > 2 else {}))   # optional LDS ring depth
torch.manual_seed(0)
for op, name in ((K.GEMM_TN, "TN"), (K.GEMM_NN, "NN"), (K.GEMM_NT, "NT")):
    for Kd in (64, 128, 192, 256, 320, 384, 448, 768):
        for M, N in ((256, 384), (256, 768), (128, 384)):
            for cs_on in ((False, True) if op == K.GEMM_TN else (False,)):
                if op == K.GEMM_TN:
                    A, B = torch.randn(Kd, M, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
                    ref = A.float().t() @ B.float()
                elif op == K.GEMM_NN:
                    A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
                    ref = A.float() @ B.float()
                else:
                    A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(N, Kd, device="cuda").to(BF16)
                    ref = A.float() @ B.float().t()
                out = torch.zeros(M, N, dtype=F32, device="cuda")
                cs = torch.zeros(M, dtype=F32, device="cuda") if cs_on else None
                K.gemm(op, A, B, out, colsum_out=cs, tune=T)
                err = (out - ref).abs()
                tiles = err.view(M // 128, 128, N // bn, bn).amax((1, 3)) / ref.abs().max()
                bad = (tiles > 1e-3).nonzero().tolist()
                print("%s K=%4d M=%3d N=%3d colsum=%d  max rel err %.2e  bad tiles %s" % (name, Kd, M, N, cs_on, tiles.max().item(), bad))
