#!/usr/bin/env python3
"""The reference's optimisation recipe at FULL c2 size through the captured step, for a few hundred optimisation steps
(My_cross_attention.py:797-844: 5 micro-batches per step, loss / 5, clip 1.0, AdamW over the two decay groups, linear warm-up
+ decay, zero_grad) on a small rotating set of synthetic batches: the loss must fall (the model memorises the batches), nothing
may turn NaN, and the first optimisation steps must agree with the same loop launched eagerly.  Train mode (dropout 0.1).
usage: python tools/train_soak.py [--opt-steps 200] [--pool 8] [--eager-check 3]"""
import argparse
import copy
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icka_amd  # noqa: E402
from icka_amd import synth  # noqa: E402
from icka_amd.config import BertConfig  # noqa: E402
from icka_amd.graph import GraphedStep  # noqa: E402
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF  # noqa: E402
from icka_amd.optim import ArenaAdamW  # noqa: E402

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")
K_ACC = 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt-steps", type=int, default=200)
    ap.add_argument("--pool", type=int, default=8)
    ap.add_argument("--eager-check", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-4)
    args = ap.parse_args()
    torch.manual_seed(synth.REFERENCE_SEED)
    cfg = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    base = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(base)
    icka_amd.set_precision(base, "bf16")
    pool = []
    for i in range(args.pool):
        b = synth.synthetic_batch(32, 128, 36, seed=500 + i)
        pool.append(tuple(b[k].cuda() for k in NAMES))

    def run(graphed, opt_steps, eval_mode):
        model = copy.deepcopy(base).cuda()
        model = model.eval() if eval_mode else model.train()

        def micro(ids, seg, mask, added, vmean, vatt, labels):
            loss = model(ids, seg, mask, added, vmean, vatt, labels=labels) / K_ACC
            loss.backward()
            return loss

        micro(*pool[0])                                  # builds the arena
        model.zero_grad()
        model._icka_arena.shadow_policy = "tracked"
        opt = ArenaAdamW(model, lr=args.lr, weight_decay=0.01, max_grad_norm=1.0)
        warm = max(1, opt_steps // 10)
        sched = torch.optim.lr_scheduler.LambdaLR(
            opt, lambda s: float(s) / warm if s < warm else max(0.0, float(opt_steps - s) / max(1, opt_steps - warm)))
        step = GraphedStep(model, micro, inputs=pool[0]) if graphed else None
        losses = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in range(opt_steps):
            acc = torch.zeros((), device="cuda")
            for m in range(K_ACC):
                b = pool[(o * K_ACC + m) % len(pool)]
                loss = step(*b) if graphed else micro(*b)
                acc += loss.detach()
            opt.step()
            sched.step()
            model.zero_grad()
            losses.append(acc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if step is not None:
            step.close()
        return [x.item() for x in losses], dt

    # (1) eval mode (dropout off): graphed == eager for the first optimisation steps
    le, _ = run(False, args.eager_check, True)
    lg, _ = run(True, args.eager_check, True)
    worst = max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(le, lg))
    print("eval mode, first %d optimisation steps: eager %s | captured %s | worst relative difference %.2e"
          % (args.eager_check, ["%.5f" % x for x in le], ["%.5f" % x for x in lg], worst), flush=True)
    assert worst < 2e-4, worst
    # (2) train mode soak through the captured step
    losses, dt = run(True, args.opt_steps, False)
    assert all(x == x and abs(x) < 1e4 for x in losses), "NaN / inf in the loss sequence"
    n = len(losses)
    first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
    print("train mode, %d optimisation steps x %d micro-batches of 32 (pool of %d batches), lr %.0e, clip 1.0, ArenaAdamW, captured step: "
          "%.1f s = %.2f ms per micro-batch incl. the update every fifth; loss (sum of the 5 scaled micro-losses) %.4f -> %.4f"
          % (n, K_ACC, args.pool, args.lr, dt, 1e3 * dt / (n * K_ACC), first, last), flush=True)
    print("loss every %d steps: %s" % (max(1, n // 20), " ".join("%.3f" % x for x in losses[::max(1, n // 20)])))
    assert last < 0.7 * first, (first, last)


if __name__ == "__main__":
    main()
