#!/usr/bin/env python3
"""mixed16 range safety probe: scale the FFN-up and gate weights of the tiny seeded model until the
oracle's activations pass 1e4 .. 6.5e4 (fp16's largest finite value is 65504) and report, per scale: the oracle's largest
GELU output and gate pre-activation, whether the mixed16 / bf16 forward and backward stay finite, and the logits error against
the oracle.  Reference sites of the scaled layers: BertIntermediate (Cross_Modal_Interaction_Module.py:548-551), the gate
(my_bert/cl_modeling.py:1363-1371)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import icka_amd  # noqa: E402
from icka_amd import synth  # noqa: E402
from oracle import mner_oracle as O  # noqa: E402


def scaled_case(scale, layers=2):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=layers, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    m = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(m)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("intermediate.dense.weight") or n.startswith("Gate_") and n.endswith("weight"):
                p.mul_(scale)
    return cfg, m


def oracle_run(m, batch):
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=128, num_hidden_layers=len(m.bert.encoder.layer), num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=64)
    peak = {"gelu": 0.0}
    real = O.gelu_erf

    def spy(x):
        y = real(x)
        peak["gelu"] = max(peak["gelu"], y.detach().abs().max().item())
        return y
    O.gelu_erf = spy
    try:
        logits = O.mner_logits(P, ocfg, batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                               batch["added_attention_mask"], batch["visual_embeds_att"], 1, 36)
    finally:
        O.gelu_erf = real
    loss = O.token_ce_loss(logits, batch["labels"], batch["input_mask"])
    loss.backward()
    return logits.detach(), loss.item(), {k: v.grad for k, v in P.items()}, peak["gelu"]


def product_run(m, batch, precision):
    model = icka_amd.set_precision(m.cuda().eval(), precision)
    g = {k: v.cuda() for k, v in batch.items()}
    model.zero_grad()
    logits = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
                   g["visual_embeds_att"])
    loss = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
                 g["visual_embeds_att"], labels=g["labels"])
    loss.backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return logits.detach().cpu(), loss.item(), grads


def main():
    import copy
    batch = synth.synthetic_batch(4, 32, 36, vocab_size=512, seed=5)
    for scale in (1, 32, 256, 1024, 4096, 16384):
        cfg, m = scaled_case(float(scale))
        ol, oloss, ograds, peak = oracle_run(m, batch)
        line = "scale %6d: oracle max|gelu out| %.3e, max|logit| %.2f, loss %.4f" % (scale, peak, ol.abs().max().item(), oloss)
        for prec in ("mixed16", "bf16"):
            pl, ploss, pg = product_run(copy.deepcopy(m), batch, prec)
            fin = bool(torch.isfinite(pl).all()) and all(bool(torch.isfinite(v).all()) for v in pg.values())
            err = (pl - ol).abs().max().item() if torch.isfinite(pl).all() else float("nan")
            gw, gk = 0.0, ""
            gmax = max(v.norm().item() for v in ograds.values() if v is not None)
            for n, v in pg.items():
                if ograds.get(n) is None or not torch.isfinite(v).all():
                    continue
                r = (v - ograds[n]).norm().item() / (ograds[n].norm().item() + 1e-3 * gmax)
                if r > gw:
                    gw, gk = r, n
            line += "\n      %-7s finite=%s  logits max abs err %.3e  loss %.4f  worst grad rel-L2 %.3e (%s)" % (prec, fin, err, ploss, gw, gk)
        print(line, flush=True)


if __name__ == "__main__":
    main()
