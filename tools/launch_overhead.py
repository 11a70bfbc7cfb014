#!/usr/bin/env python3
"""Diagnostic: cost of a kernel boundary inside a replayed hipGraph -- a chain of N dependent tiny launches (8-element
bf16 add), and the same chain with a 12.6 MB streaming add in each link (is the boundary cost hidden behind real work?)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import kernels as K  # noqa: E402

N = 230
for n_el in (8, 4096 * 768 * 2):
    a = torch.zeros(n_el, dtype=torch.bfloat16, device="cuda")
    b = torch.ones(n_el, dtype=torch.bfloat16, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            K.add_bf16(a, b, a)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(N):
                K.add_bf16(a, b, a)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 50 / N * 1e6
    print("chain of %d dependent launches, %9d elements each: %.2f us per launch (%.1f MB moved -> %.2f TB/s)"
          % (N, n_el, us, 6e-6 * n_el, 6.0 * n_el / us * 1e-6))
