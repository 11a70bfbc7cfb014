#!/usr/bin/env python3
"""Diagnostic: which ATen ops (torch-side fills / copies / adds = extra kernel launches) one eager c2 step still issues,
with the icka_amd call site of each."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from icka_amd import synth
class A: pass
args = A(); args.hidden=768; args.layers=12; args.cross_layers=1; args.labels=13; args.regions=36; args.fp8_cross=False
dev = torch.device("cuda",0)
model, cfg = bench.build_model(args, dev)
b = synth.synthetic_batch(32, 128, 36, num_labels=13)
g = {k: v.to(dev) for k, v in b.items()}
def step():
    loss = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], None, g["visual_embeds_att"], labels=g["labels"])
    loss.backward(); return loss
for _ in range(3):
    model.zero_grad(); step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    model.zero_grad(); step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.key not in ("aten::empty","aten::view","aten::as_strided","aten::empty_like","aten::empty_strided","aten::_unsafe_view","aten::reshape","aten::select","aten::slice","aten::t","aten::transpose","aten::detach","aten::alias","aten::unsqueeze","aten::expand","aten::squeeze","aten::stride","aten::is_same_size","aten::result_type","aten::to","aten::item","aten::_local_scalar_dense","aten::lift_fresh","aten::view_as")]
for e in sorted(rows, key=lambda e: -e.count)[:40]:
    st = [s for s in e.stack if "icka_amd" in s or "bench" in s or "prof_ops" in s][:2] or list(e.stack)[:4]
    print(e.key, e.count, " | ".join(s.split("/")[-1][:90] for s in st))
