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
T = K.gemm_tune(ring=ring)          # per-call word of the descriptor (no setters)
lib.icka_diag_gemm_stamp_buffer.argtypes = [__import__("ctypes").c_void_p]   # exported by -DICKA_GEMM_STAMP builds only
for name, op, M, N, Kd in (("Wo TN", K.GEMM_TN, 768, 768, 4096), ("ffndn NT", K.GEMM_NT, 4096, 768, 3072),
                           ("qkv NT", K.GEMM_NT, 4096, 2304, 768), ("outproj NT", K.GEMM_NT, 4096, 768, 768),
                           ("ffnup NT", K.GEMM_NT, 4096, 3072, 768), ("dffnup NN", K.GEMM_NN, 4096, 768, 3072)):
    if op == K.GEMM_NT:
        A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(N, Kd, device="cuda").to(BF16)
    elif op == K.GEMM_NN:
        A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
    else:
        A, B = torch.randn(Kd, M, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
    out = torch.empty(M, N, dtype=BF16, device="cuda")
    buf = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")   # >= any grid here (96-wide tiles: more blocks)
    for _ in range(20):   # warm the clocks
        K.gemm(op, A, B, out, tune=T)
    lib.icka_diag_gemm_stamp_buffer(buf.data_ptr())
    K.gemm(op, A, B, out, tune=T)
    torch.cuda.synchronize()
    lib.icka_diag_gemm_stamp_buffer(None)
    b = buf.double().cpu()
    b = b[b[:, 6] > 0]
    nb = b.shape[0]
    if nb == 0:
        print("%-10s (kernel without stamps)" % name)
        continue
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

# ---- chain mode: the same GEMMs inside a layer-like sequence (operands produced by the previous launch, not re-read
# warm from a loop over one GEMM): stamps of one GEMM of the last pass
if len(sys.argv) > 2 and sys.argv[2] == "chain":
    M = 4096
    X = torch.randn(M, 768, device="cuda").to(BF16)
    Wqkv, Wo = torch.randn(2304, 768, device="cuda").to(BF16), torch.randn(768, 768, device="cuda").to(BF16)
    W1, W2 = torch.randn(3072, 768, device="cuda").to(BF16), torch.randn(768, 3072, device="cuda").to(BF16)
    QKV, AO = torch.empty(M, 2304, dtype=BF16, device="cuda"), torch.empty(M, 768, dtype=BF16, device="cuda")
    H, Y = torch.empty(M, 3072, dtype=BF16, device="cuda"), torch.empty(M, 768, dtype=BF16, device="cuda")
    seq = (("qkv", X, Wqkv, QKV), ("outproj", QKV[:, :768], Wo, AO), ("ffnup", AO, W1, H), ("ffndn", H, W2, Y))
    buf = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")
    for target in range(4):
        buf.zero_()
        for it in range(12):
            for i, (name, a, w, o) in enumerate(seq):
                hit = it == 11 and i == target
                if hit:
                    lib.icka_diag_gemm_stamp_buffer(buf.data_ptr())
                K.gemm(K.GEMM_NT, a, w, o, tune=T)
                if hit:
                    lib.icka_diag_gemm_stamp_buffer(None)
        torch.cuda.synchronize()
        b = buf.double().cpu()
        b = b[b[:, 6] > 0]
        if b.shape[0] == 0:
            print("chain %-8s (kernel without stamps)" % seq[target][0])
            continue
        nk = b[:, 6].mean().item()
        per = b[:, :4].mean(0) / nk
        ph = b[:, 11:14].mean(0)
        clk = (b[:, 4] / b[:, 5] * 100.0).median().item()
        span = (b[:, 15].max() - b[:, 14].min()).item()
        print("chain %-8s blocks %4d nk %3d | prologue %6.0f loop %7.0f epilogue %6.0f | grid span %7.0f cyc = %.1f us | LOADER per "
              "k-tile: vmcnt-wait %5.0f barrier %5.0f issue %5.0f | clock %.0f MHz"
              % (seq[target][0], b.shape[0], nk, ph[0], ph[1], ph[2], span, span / clk, per[0], per[1], per[2], clk))
