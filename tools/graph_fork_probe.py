"""Diagnostic: what makes the captured step slower once the data-parallel machinery is on (bench.py --force-dist at world 1
was +0.75 ms)?  Replays the c2 step as a hipGraph (a) plain, (b) with N trivial side-stream fork/joins inside backward and no
process group, (c) with a process group initialised but nothing of it in the graph."""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from icka_amd import kernels as K, synth
from icka_amd.config import BertConfig
from icka_amd.graph import GraphedStep
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF

mode = sys.argv[1]
dev = torch.device("cuda", 0)
if mode == "pg":
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
cfg = BertConfig(30522)
model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
synth.fill_module_(model)
model = model.to(dev).train()
g = {k: v.to(dev) for k, v in synth.synthetic_batch(32, 128, 36).items()}
side = torch.cuda.Stream()
scratch = torch.zeros(1024, device=dev)
nfork = int(sys.argv[2]) if len(sys.argv) > 2 else 0


class Hook(object):   # stands in for GradReducer: mark_final forks a trivial kernel onto the side stream
    def __init__(self):
        self.n = 0

    def mark_final(self, slots):
        if self.n < nfork:
            self.n += 1
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                scratch.add_(1.0)

    def finish(self):
        if nfork:
            torch.cuda.current_stream().wait_stream(side)
        self.n = 0


hook = Hook()


def step():
    loss = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
                 g["visual_embeds_att"], labels=g["labels"])
    loss.backward()
    hook.finish()
    return loss


step()
model._icka_arena.reducer = hook if nfork else None
gs = GraphedStep(model, step)
for _ in range(10):
    model.zero_grad()
    gs()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    model.zero_grad()
    gs()
torch.cuda.synchronize()
print("%s forks=%d: %.3f ms/step" % (mode, nfork, 10 * (time.perf_counter() - t0)), flush=True)
if mode == "pg":
    dist.destroy_process_group()
