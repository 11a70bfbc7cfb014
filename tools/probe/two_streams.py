#!/usr/bin/env python3
"""Probe: does the c2 step gain from running as TWO half-batch chains side by side?  Two independent replicas (own weights, own
arenas) each capture a batch-16 step; the two graphs are replayed on two streams at the same time and compared with ONE batch-32
graph (and one batch-16 graph alone).  If 2 x 16 side by side beat 1 x 32, the fixed cost per kernel (ramp, first-load latency,
epilogue: ~1/3 of a short kernel) overlaps across chains and an in-GPU two-chain schedule would pay.
usage: python tools/probe/two_streams.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import icka_amd  # noqa: E402
from icka_amd import synth  # noqa: E402
from icka_amd.graph import GraphedStep  # noqa: E402

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def replica(B, seed):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    m = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(m)
    icka_amd.set_precision(m, "bf16")
    m = m.cuda().train()
    b = synth.synthetic_batch(B, 128, 36, seed=seed)
    b = tuple(b[k].cuda() for k in NAMES)

    def micro(*t):
        loss = m(*t[:6], labels=t[6])
        loss.backward()
        return loss
    m(*b[:6], labels=b[6]).backward()
    m._icka_arena.shadow_policy = "tracked"
    gs = GraphedStep(m, micro, inputs=b)
    return m, gs, b


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


m32, g32, b32 = replica(32, 1)
one32 = timed(lambda: (m32.zero_grad(), g32()), steps)
print("one chain,  batch 32          : %.3f ms per 32 samples" % one32, flush=True)
ma, ga, ba = replica(16, 2)
mb, gb, bb = replica(16, 3)
one16 = timed(lambda: (ma.zero_grad(), ga()), steps)
print("one chain,  batch 16          : %.3f ms per 16 samples (x2 = %.3f)" % (one16, 2 * one16), flush=True)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ma.zero_grad(); mb.zero_grad()


def both():
    with torch.cuda.stream(sa):
        ga()
    with torch.cuda.stream(sb):
        gb()


torch.cuda.synchronize()
two = timed(both, steps)
print("two chains, batch 16 + 16     : %.3f ms per 32 samples (side by side on two streams; gradients accumulate)" % two, flush=True)
print("ratio two chains / one chain  : %.3f" % (two / one32))
