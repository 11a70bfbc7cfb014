#!/usr/bin/env python3
"""Which hipBLASLt kernels torch.matmul picks for a few shapes (run under rocprofv3 --kernel-trace; reference point only)."""
import sys
import torch
BF16 = torch.bfloat16
shapes = [("NT", 8192, 4096, 1024), ("NN", 8192, 4096, 1024), ("NT", 8192, 3072, 1024), ("NT", 8192, 1024, 4096),
          ("NT", 4096, 2304, 768), ("NT", 4096, 3072, 768), ("NT", 4096, 768, 3072), ("NT", 4096, 768, 768)]
for op, M, N, K in shapes:
    sets = []
    for _ in range(8):
        A = torch.randn(M, K, device="cuda").to(BF16)
        B = (torch.randn(N, K, device="cuda") if op == "NT" else torch.randn(K, N, device="cuda")).to(BF16)
        sets.append((A, B, torch.empty(M, N, dtype=BF16, device="cuda")))
    torch.cuda.synchronize()
    for _ in range(3):
        for A, B, o in sets:
            torch.matmul(A, B.t() if op == "NT" else B, out=o)
    torch.cuda.synchronize()
    print(op, M, N, K, flush=True)
