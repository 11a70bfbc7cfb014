"""Diagnostic: per-layer error of the bf16 path against the fp32-exact mode (same weights, eval mode) -- how the bf16
rounding noise of the hidden states grows with depth (DESIGN.md section 2, c4 budget)."""
import copy
import sys

import torch

sys.path.insert(0, ".")
import icka_amd
from icka_amd import synth
from icka_amd.config import BertConfig
from icka_amd.modeling import BertModel

large = len(sys.argv) > 1 and sys.argv[1] == "large"
cfg = BertConfig(30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096) if large \
    else BertConfig(30522)
B, S = (4, 256) if large else (32, 128)
m = BertModel(cfg)
synth.fill_module_(m)
b = synth.synthetic_batch(B, S, 36, vocab_size=30522, seed=19260818)
ids, seg, msk = b["input_ids"].cuda(), b["segment_ids"].cuda(), b["input_mask"].cuda()
m16 = copy.deepcopy(m).cuda().eval()
m32 = icka_amd.set_precision(m.cuda().eval(), "fp32")
with torch.no_grad():
    l16, _ = m16(ids, seg, msk)
    l32, _ = m32(ids, seg, msk)
valid = msk.bool()
for i, (a, r) in enumerate(zip(l16, l32)):
    d = (a.float() - r)[valid]
    print("layer %2d  rms err %.3e  max err %.3e  (rms of hidden %.3f)" % (i, d.pow(2).mean().sqrt().item(),
                                                                        d.abs().max().item(), r[valid].pow(2).mean().sqrt().item()))
