#!/usr/bin/env python3
"""Recurrence kernels of the BiLSTM tail (icka_lstm_fwd / icka_lstm_bwd: one persistent launch over all S steps) on their
own: microseconds per launch and per time step.  usage: python tools/lstm_bench.py [--B 32 --S 128 --H 768] [--persistent 1]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib, kernels as K  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=32)
ap.add_argument("--S", type=int, default=128)
ap.add_argument("--H", type=int, default=768)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
B, S, H = args.B, args.S, args.H
lib = _lib.load()
dev = "cuda"
BF16, F32 = torch.bfloat16, torch.float32
torch.manual_seed(0)
gx = torch.randn(B * S, 8 * H, device=dev) * 0.5
whh = (torch.randn(8 * H, H, device=dev) * H ** -0.5).to(BF16)
y = torch.zeros(B * S, 2 * H, dtype=BF16, device=dev)
c_all = torch.zeros(B * S, 2 * H, device=dev)
act = torch.zeros(B * S, 8 * H, dtype=BF16, device=dev)
hprev = torch.zeros(B * S, 2 * H, dtype=BF16, device=dev)
dy = (torch.randn(B * S, 2 * H, device=dev) * 0.1).to(BF16)
whh_t = torch.empty(2 * H, 4 * H, dtype=BF16, device=dev)
K.transpose_bf16(whh, whh_t, 2, 4 * H, H)
dgates = torch.zeros(B * S, 8 * H, dtype=BF16, device=dev)
dcc = torch.zeros(2 * B, H, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / args.reps


for mode, split in ((1, 1), (1, 0), (0, 1), (0, 0)):
    flags = (0 if mode else _lib.LSTM_TICKETS) | (0 if split else _lib.LSTM_NO_BATCH_SPLIT)     # per-call flags (no setters)
    tf = timed(lambda: K.lstm_fwd(gx, whh, y, c_all, act, hprev, B, S, H, flags=flags))
    ysum = y.float().abs().sum().item()
    tb = timed(lambda: K.lstm_bwd(dy, whh_t, act, c_all, dgates, dcc, B, S, H, flags=flags))
    gsum = dgates.float().abs().sum().item()
    # arithmetic of the recurrence alone: per step and direction h_{t-1} [B, H] x W_hh^T [H, 4H] = 2 B H 4H FLOP (backward: the
    # transposed product, the same count); the floor of a step is the hand-off of h_t between the blocks of the persistent launch
    # (one flag-in-data word round trip through L2, ~1 us idle by the microarch guide's handoff table), not this arithmetic
    fl = 2.0 * 2 * B * H * 4 * H * S
    print("B %d S %d H %d handoff %s split %d | fwd %.1f us (%.2f us/step, %.1f TFLOP/s = %.4f of the bf16 peak) | bwd %.1f us (%.2f "
          "us/step, %.1f TFLOP/s) | checksums %.6e %.6e"
          % (B, S, H, mode, split, tf, tf / S, fl / tf * 1e-6, fl / tf * 1e-6 / 2500.0, tb, tb / S, fl / tb * 1e-6, ysum, gsum), flush=True)
