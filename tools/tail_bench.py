#!/usr/bin/env python3
"""Throughput of the SURVEY.md section 8(f) tail on one MI355X: the `_gate_1` tagger (trunk -> BiLSTM -> classifier ->
CRF token_mean loss; Cross_Modal_Interaction_Module.py:2383-2483) at the c2 shape, forward + backward, and the BiLSTM
and CRF layers on their own.  Not the headline metric (bench.py: the my_bert head with token-CE).
usage: python tools/tail_bench.py [--steps 20] [--warmup 5] [--no-graph]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import BertConfig, synth  # noqa: E402
from icka_amd.graph import GraphedStep  # noqa: E402
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF_gate_1  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=128)
    ap.add_argument("--regions", type=int, default=36)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--with-encoder", action="store_true",
                    help="also run the frozen ResNet-152 image encoder on 224x224 images inside the step "
                         "(My_cross_attention.py calls it once per batch): end-to-end image+sentence -> loss")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(synth.REFERENCE_SEED)
    cfg = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    model = MTCCMBertForMMTokenClassificationCRF_gate_1(cfg, num_labels=13).to(dev).train()
    b = synth.synthetic_batch(args.batch, args.seq, args.regions, num_labels=13)
    g = {k: v.to(dev) for k, v in b.items()}

    enc, images = None, None
    if args.with_encoder:
        from icka_amd.resnet import myResnet, resnet152
        net = resnet152().eval()
        synth.fill_resnet_(net)
        enc = myResnet(net.to(dev), False, dev)
        images = torch.randn(args.batch, 3, 224, 224, device=dev)
        g["added_attention_mask"] = torch.cat([torch.ones(args.batch, 49, dtype=torch.long, device=dev),
                                               g["input_mask"]], 1)

    def step():
        att = g["visual_embeds_att"]
        if enc is not None:
            _, _, att = enc(images)        # [B,2048,7,7] f32, the reference layout of visual_embeds_att
        loss = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["input_ids"], g["input_mask"],
                     g["segment_ids"], g["added_attention_mask"], visual_embeds_att=att,
                     output_mask=g["input_mask"], labels=g["labels"], mode="train")
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
    ms = timed(run, args.steps, args.warmup)
    loss = float(run().item())
    # the two tail layers on their own (eager, forward + backward)
    x = torch.randn(args.batch, args.seq, 768, device=dev).to(torch.bfloat16).requires_grad_(True)

    def lstm_only():
        out, _ = model.lstm(x)
        out.float().sum().backward()
    em = torch.randn(args.batch, args.seq, 13, device=dev, requires_grad=True)

    def crf_only():
        (-model.crf(em, g["labels"], mask=g["input_mask"].byte(), reduction="token_mean")).backward()
    lstm_ms = timed(lstm_only, args.steps, args.warmup)
    crf_ms = timed(crf_only, args.steps, args.warmup)
    # algorithmic GEMM FLOPs per sample, forward (x 3 = forward + backward): the trunk of bench.py's formula (BERT layers, one
    # cross layer, region projection) + the BiLSTM (input projection 2 S H 8H + recurrence 2 S H 8H over both directions) +
    # the [2H -> C] classifier; the CRF's O(S C^2) is negligible
    H, I, S, R, C = 768, 3072, args.seq, (49 if args.with_encoder else args.regions), 13
    fwd = 12 * (S * (8 * H * H + 4 * H * I) + 4 * S * S * H) + (S * (4 * H * H + 4 * H * I) + 4 * R * H * H + 4 * S * R * H) \
        + 2 * R * 2048 * H + 32 * S * H * H + 4 * S * H * C
    if args.with_encoder:
        fwd += 2 * 11.56e9
    tfl = 3 * fwd * args.batch / ms * 1e-9
    # the recurrence's floor is NOT arithmetic: S dependent steps per direction, each a hand-off of h_t between the blocks of
    # the persistent launch (measured ~3.7 us forward / ~5.3 us backward per step: tools/lstm_bench.py)
    print(json.dumps({"metric": "MNER samples/sec (fwd+bwd), _gate_1 tagger: trunk + BiLSTM + classifier + CRF",
                      "algorithmic_gflop_per_step": round(3 * fwd * args.batch * 1e-9, 1), "algorithmic_tflops": round(tfl, 1),
                      "frac_of_bf16_mfma_peak": round(tfl / 2500.0, 4),
                      "value": round(1e3 * args.batch / ms, 2), "unit": "samples/s", "ms_per_step": round(ms, 3),
                      "launch": mode, "loss": round(loss, 5), "n_gpus": 1, "dtype": "bf16", "data": "synthetic",
                      "config": {"workload": "bert-base + %d regions, seq %d, batch %d, train mode"
                                 % (49 if args.with_encoder else args.regions, args.seq, args.batch)},
                      "image_encoder": "resnet152 on 224x224 inside the step" if args.with_encoder else "none (region "
                      "features given)",
                      "bilstm_fwd_bwd_ms_eager": round(lstm_ms, 3), "crf_fwd_bwd_ms_eager": round(crf_ms, 3)}))


if __name__ == "__main__":
    main()
