#!/usr/bin/env python3
"""Where the HOST time of an eager (no hipGraph) c2 step goes: cProfile over a few steps, top functions by own / cumulative time.
usage: python tools/eager_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import icka_amd  # noqa: E402
from icka_amd import synth  # noqa: E402
from icka_amd.config import BertConfig  # noqa: E402
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
synth.fill_module_(model)
icka_amd.set_precision(model, "bf16")
model = model.cuda().train()
b = {k: v.cuda() for k, v in synth.synthetic_batch(32, 128, 36).items()}
one = torch.ones((), device="cuda")


def step():
    model.zero_grad()
    loss = model(b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"], b["visual_embeds_mean"],
                 b["visual_embeds_att"], labels=b["labels"])
    loss.backward(gradient=one)


for _ in range(3):
    step()
model._icka_arena.shadow_policy = "tracked"
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("eager: host issue %.3f ms/step, with GPU drain %.3f ms/step" % (1e3 * t_issue / steps, 1e3 * t_all / steps))
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):     # backward on this thread: visible to cProfile
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
