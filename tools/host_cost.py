#!/usr/bin/env python3
"""Host cost per launch of the Python -> ctypes -> HIP path (no GPU wait): a trivial kernel through the raw ctypes call, the same
through the kernels.py wrapper, a GEMM through kernels.gemm (descriptor build + validation + launch), torch.empty, and an empty
autograd Function round trip.   usage: python tools/host_cost.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

lib = _lib.load()
N = 2000
nonce = torch.zeros(2, dtype=torch.int32, device="cuda")
A = torch.randn(256, 64, device="cuda").to(torch.bfloat16)
B = torch.randn(128, 64, device="cuda").to(torch.bfloat16)
o = torch.empty(256, 128, dtype=torch.bfloat16, device="cuda")


def timeit(name, fn, n=N):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("%-60s %6.2f us per call (host)" % (name, 1e6 * dt / n), flush=True)


st = K._stream()
ptr = nonce.data_ptr()
timeit("raw ctypes launch (icka_bump_dropout_nonce, cached args)", lambda: lib.icka_bump_dropout_nonce(ptr, st))
timeit("kernels._stream()", K._stream)
timeit("torch.cuda.current_stream().cuda_stream", lambda: torch.cuda.current_stream().cuda_stream)
timeit("K.bump_dropout_nonce wrapper", lambda: K.bump_dropout_nonce(nonce))
timeit("torch.empty(4096, 768, bf16)", lambda: torch.empty(4096, 768, dtype=torch.bfloat16, device="cuda"))
timeit("K.gemm_desc only", lambda: K.gemm_desc(K.GEMM_NT, A, B, o))
timeit("K.gemm (desc + launch)", lambda: K.gemm(K.GEMM_NT, A, B, o))
d = K.gemm_desc(K.GEMM_NT, A, B, o)
import ctypes as C  # noqa: E402
timeit("raw icka_gemm with a prebuilt descriptor", lambda: lib.icka_gemm(C.byref(d), st))
x = torch.randn(4096, 768, device="cuda")
timeit("torch add_ (ATen eager op, for scale)", lambda: x.add_(1.0))


class Nop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        return a.view_as(a)

    @staticmethod
    def backward(ctx, g):
        return g


a = torch.ones(4, device="cuda", requires_grad=True)
timeit("autograd Function apply + backward (no kernels)", lambda: Nop.apply(a).backward(torch.ones(4, device="cuda")), 500)
