#!/usr/bin/env python3
"""PROBE of VERDICT r02 #3: ffn-up (+bias, GELU, pre-activation) -> ffn-down as ONE persistent launch whose 256-row stripes
hand over through counters (icka_gemm_ffn_pair) against the two separate launches, at the c2 shapes (M 4096, H 768, I 3072).

  python tools/ffn_pair_probe.py            correctness (bitwise) + timing: chains of LAYERS layer-like (pair | up, down) launches
                                            captured into one hipGraph each, cold operands per layer (12 buffer sets), HIP events
  ICKA_HIP_LIB=<stamp build> python tools/ffn_pair_probe.py stamp
                                            per-block cycle stamps of the pair kernel: phase-1 body, publish, stripe wait,
                                            phase-2 body (build: make -C icka_amd/csrc EXTRA=-DICKA_GEMM_STAMP OBJDIR=build_stamp
                                            TARGET=../libicka_hip_stamp.so)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
lib = _lib.load()
M, H, I, LAYERS = 4096, 768, 3072, 12
g = torch.Generator(device="cuda").manual_seed(1)


def rnd(*shape, scale=1.0, dtype=BF16):
    return (torch.randn(*shape, device="cuda", generator=g) * scale).to(dtype)


sets = []
for _ in range(LAYERS):
    sets.append(dict(x=rnd(M, H), W1=rnd(I, H, scale=0.03), b1=rnd(I, dtype=F32, scale=0.1), W2=rnd(H, I, scale=0.02),
                     g=torch.empty(M, I, dtype=BF16, device="cuda"), z=torch.empty(M, I, dtype=BF16, device="cuda"),
                     o=torch.empty(M, H, dtype=F32, device="cuda")))


def separate(s):
    K.gemm(K.GEMM_NT, s["x"], s["W1"], s["g"], bias=s["b1"], epilogue=K.EPI_GELU, out2=s["z"])
    K.gemm(K.GEMM_NT, s["g"], s["W2"], s["o"])


def paired(s):
    up = K.gemm_desc(K.GEMM_NT, s["x"], s["W1"], s["g"], bias=s["b1"], epilogue=K.EPI_GELU, out2=s["z"])
    down = K.gemm_desc(K.GEMM_NT, s["g"], s["W2"], s["o"])
    assert K.gemm_ffn_pair(up, down), "shape not eligible"


# ---- correctness: bitwise equal outputs, repeated launches (counters re-arm), error word clean
s0 = sets[0]
separate(s0)
torch.cuda.synchronize()
ref = (s0["g"].clone(), s0["z"].clone(), s0["o"].clone())
for it in range(5):
    for t in (s0["g"], s0["z"], s0["o"]):
        t.zero_()
    paired(s0)
    torch.cuda.synchronize()
    assert torch.equal(s0["g"], ref[0]) and torch.equal(s0["z"], ref[1]), "ffn-up outputs differ (iteration %d)" % it
    assert torch.equal(s0["o"], ref[2]), "ffn-down output differs (iteration %d): max %.3e" % (it, (s0["o"] - ref[2]).abs().max().item())
assert lib.icka_gemm_ffn_pair_error() == 0
print("pair == separate launches bit for bit (g, z, o), 5 launches, error word clean")

if len(sys.argv) > 1 and sys.argv[1] == "stamp":
    buf = torch.zeros(4096, 16, dtype=torch.int64, device="cuda")
    for _ in range(20):
        for s in sets:
            paired(s)
    lib.icka_gemm_set_stamp_buffer(buf.data_ptr())
    paired(sets[3])
    torch.cuda.synchronize()
    lib.icka_gemm_set_stamp_buffer(None)
    b = buf[:256, 8:12].double().cpu()
    if b.sum().item() == 0:
        print("(library built without -DICKA_GEMM_STAMP: no stamps)")
    else:
        m, mx = b.mean(0), b.max(0).values
        print("pair kernel, cycles per block (mean / max over 256 blocks): phase-1 body (ffn-up tile incl. epilogue) %6.0f / %6.0f | "
              "publish (drain + barrier + release + signal) %5.0f / %5.0f | stripe wait (+ acquire + barrier) %5.0f / %5.0f | "
              "phase-2 body (ffn-down tile) %6.0f / %6.0f" % (m[0], mx[0], m[1], mx[1], m[2], mx[2], m[3], mx[3]))
    sys.exit(0)


def timed(fn, reps=20):
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for s in sets:
            fn(s)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph):
        for s in sets:
            fn(s)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps / LAYERS     # us per layer-pair


for rnd_ in range(3):
    a, b = timed(separate), timed(paired)
    print("round %d: two launches %6.2f us per (ffn-up, ffn-down) | one persistent launch %6.2f us | delta %+5.2f us"
          % (rnd_, a, b, b - a))
assert lib.icka_gemm_ffn_pair_error() == 0
