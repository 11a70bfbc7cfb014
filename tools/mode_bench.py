#!/usr/bin/env python3
"""Side lines beside bench.py's headline (same model builder, same synthetic batch): the c2 step in the fp32 mode (f32-input MFMA,
the mode behind north_star's 1e-3 bar), and the FORWARD-ONLY path of the reference's test() loop (My_cross_attention.py:948-1089:
eval mode, no backward, batch 4 there; batch 32 beside it) -- eager launches and hipGraph replays (graph.GraphedModule under
no_grad).   usage: python tools/mode_bench.py [--steps 30]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import icka_amd  # noqa: E402
from icka_amd import synth  # noqa: E402
from icka_amd.graph import GraphedModule, GraphedStep  # noqa: E402

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")
PEAK = {"bf16": 2500.0, "mixed16": 2500.0, "fp32": 157.3}      # dense MFMA peaks, MI355X_MICROARCH.md


def build(precision, train):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    m = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(m)
    icka_amd.set_precision(m, precision)
    m = m.cuda()
    return m.train() if train else m.eval()


def batch(B, seed=0):
    b = synth.synthetic_batch(B, 128, 36, seed=1234 + seed)
    return tuple(b[k].cuda() for k in NAMES)


def timed(fn, steps, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    fl = bench.flops_per_sample(128, 36, 768, 3072, 12, 1, 13)
    # ---- training step, fp32 mode (and bf16 beside it on the same box)
    for precision in ("bf16", "fp32"):
        model = build(precision, True)
        b = batch(32)

        def micro(*t):
            loss = model(*t[:6], labels=t[6])
            loss.backward()
            return loss
        gs = GraphedStep(model, micro, inputs=b)

        def step():
            model.zero_grad()
            gs(*b)
        ms = timed(step, args.steps if precision != "fp32" else max(5, args.steps // 3))
        out = {"what": "c2 training step (fwd+bwd, train mode, hipGraph replay)", "precision": precision, "ms_per_step": round(ms, 3),
               "samples_per_s": round(32 / ms * 1e3, 1), "whole_step_tflops": round(3 * fl * 32 / ms * 1e-9, 1),
               "frac_of_dense_mfma_peak": round(3 * fl * 32 / ms * 1e-9 / PEAK[precision], 4), "peak_tflops": PEAK[precision]}
        print(json.dumps(out), flush=True)
        gs.close()
        del gs, model
        torch.cuda.empty_cache()
    # ---- forward only (the reference's test() loop), bf16
    for B in (4, 32):
        model = build("bf16", False)
        b = batch(B)
        with torch.no_grad():
            eager = timed(lambda: model(*b[:6]), args.steps)
            gm = GraphedModule(model, b[:6], {})
            graph = timed(lambda: gm(*b[:6]), args.steps)
            ref = model(*b[:6])
            got = gm(*b[:6])
        assert torch.equal(ref, got)
        print(json.dumps({"what": "forward only (eval mode, no_grad), logits", "batch": B, "precision": "bf16", "eager_ms": round(eager, 3),
                          "graph_ms": round(graph, 3), "samples_per_s_graph": round(B / graph * 1e3, 1),
                          "whole_forward_tflops": round(fl * B / graph * 1e-9, 1)}), flush=True)
        gm.close()
        del gm, model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
