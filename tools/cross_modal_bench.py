#!/usr/bin/env python3
"""Throughput of the reference's current (published) model on one MI355X (last row):
icka_amd.cross_modal.MTCCMBertForMMTokenClassificationCRF = bert-large-geometry text encoder -> region projection ->
text->image cross encoder -> CLIP alignment encoders -> prompt mapping networks -> prompt-spliced roberta-large-geometry
encoder -> scalar gate -> BiLSTM -> classifier -> CRF token_mean loss, forward + backward, train mode.
Not the headline metric (bench.py).   usage: python tools/cross_modal_bench.py [--batch 32] [--steps 10] [--no-graph]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import BertConfig, synth  # noqa: E402
from icka_amd.cross_modal import MTCCMBertForMMTokenClassificationCRF, PromptRobertaModel  # noqa: E402
from icka_amd.graph import GraphedStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(synth.REFERENCE_SEED)
    cfg = BertConfig(30522, hidden_size=1024, num_hidden_layers=args.layers, num_attention_heads=16, intermediate_size=4096)
    cfg_r = BertConfig(50265, hidden_size=1024, num_hidden_layers=args.layers, num_attention_heads=16,
                       intermediate_size=4096, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, None, PromptRobertaModel(cfg_r), layer_num1=1, num_labels=13)
    model = model.to(dev).train()
    nparam = sum(p.numel() for p in model.parameters())
    b = synth.synthetic_prompt_batch(args.batch, 128, num_labels=13)
    g = {k: (v if k == "offsets" else v.to(dev)) for k, v in b.items()}   # offsets stay on the host (:949 reads them)

    def step():
        loss = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["ori_input_ids"], g["ori_input_mask"],
                     g["ori_segment_ids"], g["added_attention_mask"], g["clip_features"], g["visual_embeds_mean"],
                     g["visual_embeds_att"], g["offsets"], g["output_mask"], labels=g["labels"], mode="train")
        loss.backward()
        return loss

    model.zero_grad()
    step()
    if args.no_graph:
        def run():
            model.zero_grad()
            return step()
        mode = "eager"
    else:
        gs = GraphedStep(model, step)

        def run():
            model.zero_grad()      # the reference loop drops the gradients after every step: the overwrite capture replays
            return gs()
        mode = "hipgraph"
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    loss = float(run().item())
    H, I, S, S2, R, L = 1024, 4096, 128, 178, 49, args.layers
    layer = lambda s: s * (8 * H * H + 4 * H * I) + 4 * s * s * H                      # BERT layer forward FLOPs
    cross = lambda sq, skv: sq * (4 * H * H + 4 * H * I) + 4 * skv * H * H + 4 * sq * skv * H
    fwd = L * layer(S) + L * layer(S2) + cross(S, R) + 2 * cross(1, S) + 2 * R * 2048 * H \
        + 2 * (H + 2048) * 3780 + 2 * 2 * 3780 * 5 * H + 2 * S * 8 * H * (2 * H) + 4 * S * H * 13
    print(json.dumps({"metric": "MNER samples/sec (fwd+bwd), published ICKA model: text encoder + cross encoder + CLIP "
                                "alignment + prompt networks + prompt encoder + gate + BiLSTM + CRF",
                      "value": round(1e3 * args.batch / ms, 2), "unit": "samples/s", "ms_per_step": round(ms, 3),
                      "launch": mode, "loss": round(loss, 5), "n_gpus": 1, "dtype": "bf16", "data": "synthetic",
                      "parameters_M": round(nparam / 1e6, 1),
                      "algorithmic_tflops": round(3 * fwd * args.batch / ms * 1e-9, 1),
                      "algorithmic_gflop_per_step": round(3 * fwd * args.batch * 1e-9, 1),
                      "frac_of_bf16_mfma_peak": round(3 * fwd * args.batch / ms * 1e-9 / 2500.0, 4),
                      "config": {"workload": "H1024 x %d layers text encoder (seq 128, 49 regions) + H1024 x %d layers "
                                 "prompt encoder (170 ids -> 178 positions), batch %d, train mode" % (L, L, args.batch)}}))


if __name__ == "__main__":
    main()
