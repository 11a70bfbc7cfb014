#!/usr/bin/env python3
"""Diagnostic: where a block of the whole-head attention backward spends its cycles (s_memtime stamps of lane 0 of
every wave).  Needs a stamp build:
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DICKA_ATTN_STAMP -c attention.hip -o build/stamp/attention.o  (+ link)
  ICKA_HIP_LIB=icka_amd/csrc/build/stamp/libicka_hip.so python tools/attn_stamp.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
lib = _lib.load()
setbuf = lib.icka_diag_attn_stamp_buffer
setbuf.argtypes = [C.c_void_p]
setbuf.restype = None
CASES = [(32, 12, 128, 128, 0.1), (32, 12, 128, 128, 0.0), (32, 12, 128, 36, 0.1),
         (32, 16, 256, 256, 0.1), (32, 16, 256, 256, 0.0)]          # the last two: bert-large at seq 256 (c4)
for B, h, Sq, Skv, p in CASES:
    H = 64 * h
    qkv = torch.randn(B * Sq, 3 * H, device="cuda").to(BF16)
    kv = torch.randn(B * Skv, 2 * H, device="cuda").to(BF16) if Skv != Sq else qkv[:, H:]
    q, k, v = qkv[:, :H], kv[:, :H], kv[:, H:2 * H]
    mask = torch.zeros(B, Skv, dtype=F32, device="cuda")
    out = torch.empty(B * Sq, H, dtype=BF16, device="cuda")
    lse = torch.empty(B, h, Sq, dtype=F32, device="cuda")
    dout = torch.randn(B * Sq, H, device="cuda").to(BF16)
    dqkv = torch.empty(B * Sq, 3 * H, dtype=BF16, device="cuda")
    dkv = torch.empty(B * Skv, 2 * H, dtype=BF16, device="cuda") if Skv != Sq else dqkv[:, H:]
    delta = torch.empty(B, h, Sq, dtype=F32, device="cuda")
    kb = K.attn_keepbits(B, h, Sq, Skv, q.device) if (p > 0 and Skv > 128) else None     # as ops.py does beyond 128 keys
    K.attn_fwd(q, k, v, mask, out, lse, B, h, Sq, Skv, p_drop=p, seed=1, keepbits=kb)
    buf = torch.zeros(B * h, 4, 16, dtype=torch.int64, device="cuda")

    def bwd():
        K.attn_bwd(q, k, v, mask, out, dout, lse, delta, dqkv[:, :H], dkv[:, :H], dkv[:, H:2 * H], B, h, Sq, Skv,
                   p_drop=p, seed=1, keepbits=kb)
    for _ in range(20):
        bwd()
    setbuf(buf.data_ptr())
    bwd()
    torch.cuda.synchronize()
    setbuf(None)
    t = buf.double().cpu()
    d = (t[:, :, 1:8] - t[:, :, :7]).mean((0, 1))
    span = (t[:, :, 9].max() - t[:, :, 8].min()).item() * 10.0   # s_memrealtime: 100 MHz -> ns
    per_block = (t[:, :, 7].amax(1) - t[:, :, 0].amin(1)).mean().item()
    blk_ns = ((t[:, :, 9].amax(1) - t[:, :, 8].amin(1)) * 10.0)
    print("Sq %d Skv %d p %.1f | cycles (s_memtime ticks, 100 MHz? see gemm_stamp) mean over waves: stage-issue %.0f  sync %.0f  "
          "phaseA %.0f  sync+write+sync %.0f  dV %.0f  sync+write+sync %.0f  dK %.0f | block mean %.0f"
          % (Sq, Skv, p, d[0], d[1], d[2], d[3], d[4], d[5], d[6], per_block))
    starts = (t[:, 0, 8] - t[:, :, 8].min()) * 10.0
    print("   realtime: grid span %.0f ns | block duration ns: mean %.0f  max %.0f | block start offset ns: median %.0f  p90 %.0f  max %.0f"
          " | implied clock %.2f GHz" % (span, blk_ns.mean().item(), blk_ns.max().item(), starts.median().item(),
                                      starts.quantile(0.9).item(), starts.max().item(), per_block / blk_ns.mean().item()))
