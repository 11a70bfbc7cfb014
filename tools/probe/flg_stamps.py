#!/usr/bin/env python3
"""Stamps of the LDS-counter hand-over form of the 128x96 GEMM kernel (diagnostic library built from commit d01c66e's gemm.hip +
spin counters, ICKA_HIP_LIB=...): per k-tile, loader wave segments (wait-for-landed, signal, wait-for-free-slot, issue), spins of
both roles, compute loop.  Warm = same operands every launch; cold = the activation operand rotates over 12 buffers."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16 = torch.bfloat16
lib = _lib.load()
raw = ctypes.CDLL(os.environ["ICKA_HIP_LIB"])
lib.icka_gemm_set_tile_n(96)
for name, op, M, N, Kd in (("ffndn NT", K.GEMM_NT, 4096, 768, 3072), ("dffnup NN", K.GEMM_NN, 4096, 768, 3072)):
    As = [torch.randn(M, Kd, device="cuda").to(BF16) for _ in range(12)]
    B = (torch.randn(N, Kd, device="cuda") if op == K.GEMM_NT else torch.randn(Kd, N, device="cuda")).to(BF16)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    buf = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")
    for cold in (0, 1):
        for fs in (0, 1, 4):
            raw.icka_gemm_set_flag_sync(fs)
            for i in range(24):
                K.gemm(op, As[i % 12 if cold else 0], B, out)
            buf.zero_()
            lib.icka_gemm_set_stamp_buffer(buf.data_ptr())
            K.gemm(op, As[0], B, out)
            torch.cuda.synchronize()
            lib.icka_gemm_set_stamp_buffer(None)
            b = buf.double().cpu()
            b = b[b[:, 6] > 0]
            nk = b[:, 6].mean().item()
            per = b[:, :4].mean(0) / nk
            ph = b[:, 11:14].mean(0)
            clk = (b[:, 4] / b[:, 5] * 100.0).median().item()
            print("%-10s %s flagsync %d | compute loop %4.0f cycles / k-tile (%3.0f ns) | loader: landed-wait %4.0f signal/barrier %4.0f "
                  "slot-wait %4.0f issue %4.0f | spins per block: loader %5.1f compute %5.1f | %4.0f MHz"
                  % (name, "cold A" if cold else "warm  ", fs, ph[1] / nk, ph[1] / nk / clk * 1e3, per[0], per[1], per[3], per[2],
                     b[:, 7].mean().item(), b[:, 8].mean().item(), clk), flush=True)
raw.icka_gemm_set_flag_sync(0)
