#!/usr/bin/env python3
"""Diagnostic: which operand roundings of the bf16 path make up its logit error.  The fp32-exact mode runs the full gated
head with selected GEMM operands (and the q/k/v projection output) rounded to bf16 or fp16 on the way in; the logits are
compared with the unrounded fp32 run.  usage: rounding_attribution.py [c4|c2]"""
import sys

import torch

sys.path.insert(0, ".")
import icka_amd
from icka_amd import exact as X
from icka_amd import synth
from icka_amd.config import BertConfig
from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF

which = sys.argv[1] if len(sys.argv) > 1 else "c4"
if which == "c4":
    cfg = BertConfig(30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
    B, S, R = 4, 256, 50
else:
    cfg = BertConfig(30522)
    B, S, R = 8, 128, 36
H, I = cfg.hidden_size, cfg.intermediate_size
model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=R)
synth.fill_module_(model)
model = icka_amd.set_precision(model.cuda().eval(), "fp32")
b = synth.synthetic_batch(B, S, R, num_labels=13, seed=7)
g = {k: v.cuda() for k, v in b.items()}
orig = X.gemm
policy = {}


def rnd(t, how):
    if how == "bf16":
        return t.to(torch.bfloat16).float()
    if how == "fp16":
        return t.to(torch.float16).float()
    return t


def patched(op, A, Bm, out, **kw):
    shp = tuple(Bm.shape)
    cat = ("ffn_up" if shp == (I, H) else "ffn_down" if shp == (H, I) else "qkv3" if shp == (3 * H, H) else
           "hh" if shp == (H, H) else "rest")
    how = policy.get(cat, policy.get("ffn" if cat.startswith("ffn") else "other"))
    if op == X.GEMM_NT and how and (how != "none" or policy.get("qkv_out")):
        A2 = rnd(A, how) if A.is_contiguous() else rnd(A.contiguous(), how)
        if not A.is_contiguous():   # keep the caller's strides simple: contiguous copy is a valid 2-D row-major view
            pass
        r = orig(op, A2, rnd(Bm, how), out, **kw)
        if policy.get("qkv_out") and shp[1] == H and shp[0] in (2 * H, 3 * H):
            out.copy_(rnd(out, policy["qkv_out"]))
        return r
    return orig(op, A, Bm, out, **kw)


def run():
    with torch.no_grad():
        return model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"],
                     g["visual_embeds_mean"], g["visual_embeds_att"]).float()


orig_ln = X.ln_fwd


def patched_ln(x, residual, gamma, beta, **kw):
    y, xhat, rstd = orig_ln(x, residual, gamma, beta, **kw)
    if policy.get("ln_out"):
        y.copy_(rnd(y, policy["ln_out"]))     # the LayerNorm output IS the residual of the next block
    return y, xhat, rstd


ref = run()
X.gemm = patched
X.ln_fwd = patched_ln
valid = g["input_mask"].bool()
for name, pol in (("all operands bf16 (+ q/k/v outputs bf16)", {"ffn": "bf16", "other": "bf16", "qkv_out": "bf16"}),
                  ("FFN operands bf16 only", {"ffn": "bf16"}),
                  ("non-FFN operands bf16 (+ q/k/v outputs) only", {"other": "bf16", "qkv_out": "bf16"}),
                  ("FFN fp16, rest bf16 (+ q/k/v outputs bf16)", {"ffn": "fp16", "other": "bf16", "qkv_out": "bf16"}),
                  ("all operands fp16 (+ q/k/v outputs fp16)", {"ffn": "fp16", "other": "fp16", "qkv_out": "fp16"}),
                  ("only ffn-up operands bf16", {"ffn_up": "bf16"}), ("only ffn-down operands bf16", {"ffn_down": "bf16"}),
                  ("only fused qkv operands bf16", {"qkv3": "bf16"}), ("only q/k/v outputs bf16", {"qkv3": "none", "qkv_out": "bf16"}),
                  ("only [H,H] operands bf16 (out-proj, gates, cross q)", {"hh": "bf16"}),
                  ("only the rest bf16 (vismap2text, cross k/v, classifier)", {"rest": "bf16"}),
                  ("FFN + qkv operands fp16, rest bf16, q/k/v out bf16", {"ffn": "fp16", "qkv3": "fp16", "other": "bf16", "qkv_out": "bf16"}),
                  ("FFN + qkv + [H,H] fp16, rest bf16, q/k/v out bf16", {"ffn": "fp16", "qkv3": "fp16", "hh": "fp16", "other": "bf16", "qkv_out": "bf16"}),
                  ("everything fp16 but q/k/v outputs bf16", {"ffn": "fp16", "other": "fp16", "qkv_out": "bf16"}),
                  ("same + LayerNorm outputs (residual stream) fp16", {"ffn": "fp16", "other": "fp16", "qkv_out": "bf16", "ln_out": "fp16"}),
                  ("layers fp16 + fp16 residual; rest (head, regions) bf16", {"ffn": "fp16", "qkv3": "fp16", "hh": "fp16", "other": "bf16", "qkv_out": "bf16", "ln_out": "fp16"}),
                  ("all bf16 + LayerNorm outputs (residual stream) bf16", {"ffn": "bf16", "other": "bf16", "qkv_out": "bf16", "ln_out": "bf16"})):
    policy.clear()
    policy.update(pol)
    d = (run() - ref)[valid]
    print("%-50s max |dlogit| %.3e   rms %.3e" % (name, d.abs().max().item(), d.pow(2).mean().sqrt().item()), flush=True)
X.gemm = orig
X.ln_fwd = orig_ln
