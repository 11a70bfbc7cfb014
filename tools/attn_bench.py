#!/usr/bin/env python3
"""Attention kernels at the c2 shapes (self 128x128, text->image 128x36), to be run under
`rocprofv3 --kernel-trace --stats` (per-kernel durations); ICKA_HIP_LIB selects a diagnostic build
(-DICKA_ATTN_ABLATE=1: no global loads, =2: staging + stores only).
usage: python tools/attn_bench.py [--iters 20] [--p 0.1] [--whole 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--p", type=float, default=0.1)
    ap.add_argument("--whole", type=int, default=1)
    ap.add_argument("--c4", action="store_true", help="bert-large geometry: 16 heads, 256 x 256 (tiled kernels)")
    ap.add_argument("--shape", default=None, help="Sq,Skv,heads: one explicit shape (batch 32)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--keepbits", type=int, default=0, help="1: the forward leaves its keep bits, the backward reads them")
    args = ap.parse_args()
    tiled = not bool(args.whole)      # a per-call flag of icka_attn_fwd_ex / icka_attn_bwd (no process-wide switch)
    B, h, H = (args.batch, 16, 1024) if args.c4 else (args.batch, 12, 768)
    torch.manual_seed(0)
    shapes = ((256, 256), (256, 50)) if args.c4 else ((128, 128), (128, 36))
    if args.shape:
        sq, skv, h = (int(v) for v in args.shape.split(","))
        H, shapes = 64 * h, ((sq, skv),)
    for Sq, Skv in shapes:
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
        kb = K.attn_keepbits(B, h, Sq, Skv, "cuda") if (args.keepbits and args.p > 0) else None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        for it in range(args.iters + 3):
            if it == 3:
                ev[0].record()
            K.attn_fwd(q, k, v, mask, out, lse, B, h, Sq, Skv, p_drop=args.p, seed=1234, keepbits=kb, tiled=tiled)
        ev[1].record()
        for it in range(args.iters):
            K.attn_bwd(q, k, v, mask, out, dout, lse, delta, dqkv[:, :H], dkv[:, :H], dkv[:, H:2 * H], B, h, Sq, Skv,
                       p_drop=args.p, seed=1234, keepbits=kb, tiled=tiled)
        ev[2].record()
        torch.cuda.synchronize()
        print("B %d heads %d Sq %d Skv %d p %.2f whole %d keepbits %d: fwd %.1f us, bwd %.1f us (HIP events, back-to-back launches)"
              % (B, h, Sq, Skv, args.p, args.whole, args.keepbits, 1e3 * ev[0].elapsed_time(ev[1]) / args.iters,
                 1e3 * ev[1].elapsed_time(ev[2]) / args.iters))


if __name__ == "__main__":
    main()
