#!/usr/bin/env python3
"""Throughput of the frozen ResNet-152 image encoder (resnet/resnet_utils.py myResnet.forward,
called once per batch at My_cross_attention.py) on one MI355X: 32 images of 224x224, forward only.
usage: python tools/resnet_bench.py [--batch 32] [--steps 20] [--no-graph]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import synth  # noqa: E402
from icka_amd.resnet import myResnet, resnet152  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    net = resnet152().eval()
    synth.fill_resnet_(net)
    enc = myResnet(net.cuda(), False, None)
    x = torch.randn(args.batch, 3, 224, 224, device="cuda")
    for _ in range(args.warmup):
        out = enc(x)
    torch.cuda.synchronize()
    mode = "eager"
    run = lambda: enc(x)
    if not args.no_graph:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            enc(x)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            out = enc(x)
        run = g.replay
        mode = "hipgraph"
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    flops = 2 * 11.56e9 * args.batch     # 11.56 G multiply-adds per 224x224 image (ResNet-152)
    print(json.dumps({"metric": "ResNet-152 image encoder, forward (frozen), images/s", "value": round(1e3 * args.batch / ms, 1),
                      "unit": "images/s", "ms_per_batch": round(ms, 3), "batch": args.batch, "launch": mode,
                      "algorithmic_tflops": round(flops / ms / 1e9, 1), "algorithmic_gflop_per_batch": round(flops * 1e-9, 1),
                      "frac_of_bf16_mfma_peak": round(flops / ms / 1e9 / 2500.0, 4), "dtype": "bf16", "n_gpus": 1}))


if __name__ == "__main__":
    main()
