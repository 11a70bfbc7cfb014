#!/usr/bin/env python3
"""Fixed-vs-per-iteration cost probe of the GEMM fast path: time(K) for a few K at fixed M,N."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2]


def main():
    ring = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    T = K.gemm_tune(ring=ring)      # per-call word of the descriptor (no setter)
    print("ring", ring)
    for op, name in ((K.GEMM_NT, "NT"), (K.GEMM_TN, "TN")):
        for (M, N, odt) in ((4096, 768, BF16), (4096, 768, F32), (4096, 3072, BF16), (4096, 2304, BF16)):
            row = []
            for Kd in (64, 128, 256, 768, 1536, 3072):
                if op == K.GEMM_NT:
                    A, B = torch.randn(M, Kd, device="cuda").to(BF16), torch.randn(N, Kd, device="cuda").to(BF16)
                else:
                    A, B = torch.randn(Kd, M, device="cuda").to(BF16), torch.randn(Kd, N, device="cuda").to(BF16)
                out = torch.empty(M, N, dtype=odt, device="cuda")
                t = timeit(lambda: K.gemm(op, A, B, out, tune=T))
                row.append("K=%d: %.1fus" % (Kd, t))
            print("%s M=%d N=%d out=%s | %s" % (name, M, N, "bf16" if odt == BF16 else "f32", "  ".join(row)))
    # launch floor: an (almost) empty kernel
    x = torch.zeros(64, 8, dtype=BF16, device="cuda"); y = torch.empty_like(x)
    print("tiny dropout kernel: %.1f us" % timeit(lambda: K.dropout(x, y, p_drop=0.0, seed=0)))


if __name__ == "__main__":
    main()
