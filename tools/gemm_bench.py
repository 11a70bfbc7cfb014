#!/usr/bin/env python3
"""Per-shape timing of the GEMM launches the c2 workload issues (HIP events, interleaved rounds, random data).
usage: python tools/gemm_bench.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
M, H, I, R = 4096, 768, 3072, 1152

# (name, op, M, N, K, epilogue, out dtype)   shapes of one BERT layer fwd+bwd at c2
SHAPES = [
    ("fwd qkv      NT", K.GEMM_NT, M, 3 * H, H, K.EPI_NONE, BF16),
    ("fwd out-proj NT", K.GEMM_NT, M, H, H, K.EPI_NONE, F32),
    ("fwd ffn-up   NT", K.GEMM_NT, M, I, H, K.EPI_GELU, BF16),
    ("fwd ffn-down NT", K.GEMM_NT, M, H, I, K.EPI_NONE, F32),
    ("bwd d-ffn-dn NN", K.GEMM_NN, M, I, H, K.EPI_DGELU, BF16),
    ("bwd d-ffn-up NN", K.GEMM_NN, M, H, I, K.EPI_ADD, BF16),
    ("bwd d-out    NN", K.GEMM_NN, M, H, H, K.EPI_NONE, BF16),
    ("bwd d-qkv    NN", K.GEMM_NN, M, H, 3 * H, K.EPI_ADD, BF16),
    ("wgrad W2     TN", K.GEMM_TN, H, I, M, K.EPI_NONE, F32),
    ("wgrad W1     TN", K.GEMM_TN, I, H, M, K.EPI_NONE, F32),
    ("wgrad Wo     TN", K.GEMM_TN, H, H, M, K.EPI_NONE, F32),
    ("wgrad Wqkv   TN", K.GEMM_TN, 3 * H, H, M, K.EPI_NONE, F32),
    ("vismap2text  NT", K.GEMM_NT, R, H, 2048, K.EPI_NONE, BF16),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--ring", type=int, default=3)
    ap.add_argument("--ablate", type=int, default=0)
    ap.add_argument("--ws", type=int, default=1)
    args = ap.parse_args()
    T0 = K.gemm_tune(ring=args.ring, warp_specialized=args.ws)      # per-call words of the descriptor (no setters)
    T = K.gemm_tune(ring=args.ring, warp_specialized=args.ws, **({"ablation": args.ablate} if args.ablate else {}))
    print("ring depth", args.ring, "ablation", args.ablate, "warp-specialised", args.ws)
    torch.manual_seed(0)
    cases = []
    for name, op, m, n, k, epi, odt in SHAPES:
        if op == K.GEMM_NT:
            A, B = torch.randn(m, k, device="cuda").to(BF16), torch.randn(n, k, device="cuda").to(BF16)
            ref = A.float() @ B.float().t()
        elif op == K.GEMM_NN:
            A, B = torch.randn(m, k, device="cuda").to(BF16), torch.randn(k, n, device="cuda").to(BF16)
            ref = A.float() @ B.float()
        else:
            A, B = torch.randn(k, m, device="cuda").to(BF16), torch.randn(k, n, device="cuda").to(BF16)
            ref = A.float().t() @ B.float()
        out = torch.empty(m, n, dtype=odt, device="cuda")
        aux = torch.randn(m, n, device="cuda").to(BF16) if epi in (K.EPI_DGELU, K.EPI_ADD) else None
        out2 = torch.empty(m, n, dtype=BF16, device="cuda") if epi == K.EPI_GELU else None
        bias = torch.randn(n, device="cuda") if epi == K.EPI_GELU else None
        K.gemm(op, A, B, out, epilogue=K.EPI_NONE, tune=T0)
        err = ((out.float() - ref).abs().max() / ref.abs().max()).item()
        cases.append((name, op, m, n, k, epi, A, B, out, aux, out2, bias, err))
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in cases]
          for _ in range(args.iters)]
    for it in range(args.iters + 3):
        for ci, (name, op, m, n, k, epi, A, B, out, aux, out2, bias, err) in enumerate(cases):
            if it >= 3:
                ev[it - 3][ci][0].record()
            K.gemm(op, A, B, out, epilogue=epi, aux=aux, out2=out2, bias=bias, tune=T)
            if it >= 3:
                ev[it - 3][ci][1].record()
    torch.cuda.synchronize()
    tot_us, tot_fl = 0.0, 0.0
    for ci, c in enumerate(cases):
        ts = sorted(ev[it][ci][0].elapsed_time(ev[it][ci][1]) * 1e3 for it in range(args.iters))
        med, mn = ts[len(ts) // 2], ts[0]
        fl = 2.0 * c[2] * c[3] * c[4]
        tot_us += med
        tot_fl += fl
        print("%-18s M=%5d N=%5d K=%5d  med %7.1f us  min %7.1f us  %7.1f TF/s  (rel err %.1e)"
              % (c[0], c[2], c[3], c[4], med, mn, fl / med / 1e6, c[12]))
    print("layer-sum: %.1f us, %.1f TF/s" % (tot_us, tot_fl / tot_us / 1e6))


if __name__ == "__main__":
    main()
