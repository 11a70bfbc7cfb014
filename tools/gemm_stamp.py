#!/usr/bin/env python3
"""Diagnostic (build with `make EXTRA=-DICKA_GEMM_STAMP`): where a k-loop iteration of the fast-path GEMM spends its
cycles (s_memtime stamps of wave 0 of every block) and the in-kernel clock (s_memtime / s_memrealtime)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
lib = _lib.load()
ring = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib.icka_gemm_set_ring(ring)
for name, op, M, N, Kd in (("Wo TN", K.GEMM_TN, 768, 768, 4096), ("ffndn NT", K.GEMM_NT, 4096, 768, 3072),
                           ("qkv NT", K.GEMM_NT, 4096, 2304, 768), ("outproj NT", K.GEMM_NT, 4096, 768, 768),
                           ("ffnup NT", K.GEMM_NT, 4096, 3072, 768), ("dffnup NN", K.GEMM_NN, 4096, 768, 3072)):
    if op == K.GEMM_NT:
        A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(N, Kd, device="cuda").to(BF16)
    elif op == K.GEMM_NN:
        A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
    else:
        A, B = torch.randn(Kd, M, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
    out = torch.empty(M, N, dtype=F32, device="cuda")
    nb = (M // 128) * (N // 128)
    buf = torch.zeros(nb, 16, dtype=torch.int64, device="cuda")
    for _ in range(20):   # warm the clocks
        K.gemm(op, A, B, out)
    lib.icka_gemm_set_stamp_buffer(buf.data_ptr())
    K.gemm(op, A, B, out)
    torch.cuda.synchronize()
    lib.icka_gemm_set_stamp_buffer(None)
    b = buf.double().cpu()
    nk = b[:, 6].mean().item()
    tot = b[:, 4]
    clk = (b[:, 4] / b[:, 5] * 100.0).median().item()   # MHz
    per = b[:, :4].mean(0) / nk
    ph = b[:, 11:14].mean(0)
    span = (b[:, 15].max() - b[:, 14].min()).item()
    print("%-10s phases (cycles, mean over blocks): prologue %6.0f  loop %7.0f  epilogue %6.0f | whole grid span %8.0f cycles, "
          "sum of phases %7.0f, rounds %.2f" % (name, ph[0], ph[1], ph[2], span, ph.sum().item(), nb / 256.0))
    cper = b[:, 8:10].mean(0) / nk
    print("%-10s ring %d blocks %4d nk %3d | LOADER per k-tile: vmcnt-wait %5.0f barrier %5.0f dma-issue %5.0f (loop %7.0f cyc) | "
          "COMPUTE per k-tile: F1-reads+MFMA(F0) %5.0f barrier-wait %5.0f (loop %7.0f cyc) | clock %.0f MHz"
          % (name, ring, nb, nk, per[0], per[1], per[2], tot.mean().item(), cper[0], cper[1], b[:, 10].mean().item(), clk))
