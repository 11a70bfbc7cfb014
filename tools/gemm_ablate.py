#!/usr/bin/env python3
"""Diagnostic (library built with -DICKA_GEMM_STAMP -DICKA_GEMM_ABLATE, selected by ICKA_HIP_LIB): loop cycles per k-tile of
the 128x96-tile kernel with the compute side (abl 1) or the DMA staging (abl 2) removed, or with both sides complete but the
per-k-tile barrier removed (abl 3: the contention floor of any synchronisation scheme).
build: make -C icka_amd/csrc EXTRA="-DICKA_GEMM_STAMP -DICKA_GEMM_ABLATE" OBJDIR=build_diag TARGET=../libicka_hip_diag.so
run:   ICKA_HIP_LIB=icka_amd/libicka_hip_diag.so python tools/gemm_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16 = torch.bfloat16
lib = _lib.load()
lib.icka_diag_gemm_stamp_buffer.argtypes = [__import__("ctypes").c_void_p]   # exported by -DICKA_GEMM_STAMP builds only
for name, op, M, N, Kd in (("ffndn NT", K.GEMM_NT, 4096, 768, 3072), ("dffnup NN", K.GEMM_NN, 4096, 768, 3072)):
    A = torch.randn(M, Kd, device="cuda").to(BF16)
    B = (torch.randn(N, Kd, device="cuda") if op == K.GEMM_NT else torch.randn(Kd, N, device="cuda")).to(BF16)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    buf = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")
    for nlw in (4,):
        for abl in (0, 1, 2, 3):   # 3: both roles with all their work but no barriers (free-running, garbage results)
            T = K.gemm_tune(tile_n=96, **({"ablation": abl} if abl else {}))   # per-call word: no process-wide switch
            for _ in range(20):
                K.gemm(op, A, B, out, tune=T)
            buf.zero_()
            lib.icka_diag_gemm_stamp_buffer(buf.data_ptr())
            K.gemm(op, A, B, out, tune=T)
            torch.cuda.synchronize()
            lib.icka_diag_gemm_stamp_buffer(None)
            b = buf.double().cpu()
            b = b[b[:, 6] > 0]
            nk = b[:, 6].mean().item()
            per = b[:, :3].mean(0) / nk
            ph = b[:, 11:14].mean(0)
            clk = (b[:, 4] / b[:, 5] * 100.0).median().item()
            print("%-10s loaders %d abl %d | compute-wave phases: prologue %5.0f loop %6.0f (%4.0f / k-tile) epilogue %5.0f | loader per "
                  "k-tile: wait %4.0f barrier %4.0f issue %4.0f | %4.0f MHz" % (name, nlw, abl, ph[0], ph[1], ph[1] / nk, ph[2],
                                                                             per[0], per[1], per[2], clk), flush=True)
