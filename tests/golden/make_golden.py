#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself (dev container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (buctcurry/ICKA at /root/reference, read-only) is imported in-process with stub modules for
third-party packages its hot-path classes never call (SURVEY.md section 8c recipe).  For every case we
  1. build the reference module(s), overwrite every parameter with ``icka_amd.synth.seeded_tensor(key)``,
  2. run the reference forward (eval mode, dropout off) and backward of the benchmark loss,
  3. assert the CPU oracle (oracle/mner_oracle.py) reproduces the reference to <= 1e-5, and
  4. write inputs' seeds + expected outputs as a small .npz.
Only data (inputs / expected outputs) is written; no reference source travels.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from icka_amd import synth  # noqa: E402
from oracle import mner_oracle as O  # noqa: E402


def _import_reference():
    import transformers  # noqa: F401  (let accelerate's availability probes run before the stubs exist)
    from transformers import BertModel as _unused  # noqa: F401

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m

    class CRF(nn.Module):  # stands in for pytorch-crf (absent; outside the hot path): records the emissions
        def __init__(self, num_tags, batch_first=False):
            super().__init__()
            self.emissions = None

        def forward(self, emissions, tags=None, mask=None, reduction="mean"):
            self.emissions = emissions
            return emissions.sum() * 0.0

        def decode(self, emissions, mask=None):
            self.emissions = emissions
            return emissions.argmax(-1).tolist()

    class Sparsemax(nn.Module):
        def __init__(self, dim=-1):
            super().__init__()

    stub("torchcrf", CRF=CRF)
    stub("sparsemax", Sparsemax=Sparsemax)
    stub("boto3")
    stub("botocore")
    stub("botocore.exceptions", ClientError=Exception)
    sys.path.insert(0, REF)
    import Cross_Modal_Interaction_Module as CM
    import my_bert.cl_modeling as CL
    import my_bert.gate_cl_modeling as GCL
    return CM, CL, GCL


def _cfg_pair(mod, **kw):
    ref_cfg = mod.BertConfig(kw.pop("vocab_size"), **kw)
    ocfg = O.OracleConfig(vocab_size=ref_cfg.vocab_size, hidden_size=ref_cfg.hidden_size,
                          num_hidden_layers=ref_cfg.num_hidden_layers,
                          num_attention_heads=ref_cfg.num_attention_heads,
                          intermediate_size=ref_cfg.intermediate_size,
                          max_position_embeddings=ref_cfg.max_position_embeddings,
                          type_vocab_size=ref_cfg.type_vocab_size)
    return ref_cfg, ocfg


def _sample(t: torch.Tensor, n: int = 64) -> np.ndarray:
    """Deterministic strided sample of a tensor's flattened values."""
    f = t.detach().reshape(-1)
    m = min(n, f.numel())
    idx = (torch.arange(m, dtype=torch.long) * (f.numel() - 1)) // max(m - 1, 1)
    return f[idx].numpy().copy()


def _ref_compose(model, batch, regions, variant="cl"):
    """Run the reference's OWN sub-modules in the order of cl_modeling.py:1341-1371 (or gate_cl_modeling.py
    :1322-1381) for an arbitrary region count (the model's forward hard-codes 49)."""
    seq, pooled = model.bert(batch["input_ids"], token_type_ids=batch["segment_ids"],
                             attention_mask=batch["input_mask"], output_all_encoded_layers=False)
    seq = model.dropout(seq)
    vis = batch["visual_embeds_att"]
    if vis.dim() == 4:
        vis = vis.view(-1, 2048, regions).permute(0, 2, 1)
    conv = model.vismap2text(vis)
    img_mask = batch["added_attention_mask"][:, :regions][:, None, None, :].to(torch.float32)
    img_mask = (1.0 - img_mask) * -10000.0
    cross = model.txt2img_attention(seq, conv, img_mask)[-1]
    crs = None
    cross_g = cross
    if variant == "gate_cl":
        crs = model.crs_classifier(torch.cat((seq, cross), dim=-1).view(seq.shape[0], -1))
        p = torch.softmax(crs, dim=-1)[:, -1][:, None, None]
        cross_g = p * cross
    gate = torch.sigmoid(model.Gate_text(seq) + model.Gate_image(cross_g))
    logits = model.classifier(torch.cat((seq, gate * cross_g), dim=-1))
    return {"seq": seq, "vis": conv, "cross": cross, "gate": gate, "logits": logits, "pooled": pooled, "crs": crs}


def _check(name, a, b, tol=1e-5):
    err = (a.detach() - b.detach()).abs().max().item()
    assert err <= tol, "%s: oracle vs reference max abs diff %.3e > %.1e" % (name, err, tol)
    return err


def make_case(CL, GCL, name, cfg_kw, batch_kw, layer_num1, num_labels, variant="cl", full=False,
              special_masks=False):
    mod = GCL if variant == "gate_cl" else CL
    ref_cfg, ocfg = _cfg_pair(mod, **dict(cfg_kw))
    torch.manual_seed(0)
    model = mod.MTCCMBertForMMTokenClassificationCRF(ref_cfg, layer_num1=layer_num1, num_labels=num_labels)
    synth.fill_module_(model)
    model.eval()
    regions = batch_kw["regions"]
    batch = synth.synthetic_batch(num_labels=num_labels, vocab_size=ref_cfg.vocab_size, **batch_kw)
    if special_masks:
        # edge cases: sample 0 all-valid, sample 1 length 1, last sample fully masked (all-zero mask)
        s = batch["input_mask"].shape[1]
        batch["input_mask"][0] = 1
        batch["input_mask"][1] = 0
        batch["input_mask"][1, 0] = 1
        batch["input_mask"][-1] = 0
        batch["added_attention_mask"][:, regions:] = batch["input_mask"]
        # also mask out a few regions for sample 0 to exercise the region mask
        batch["added_attention_mask"][0, regions - 5:regions] = 0
        g = torch.Generator().manual_seed(7)
        batch["input_ids"] = torch.randint(1, ref_cfg.vocab_size, batch["input_ids"].shape, generator=g) \
            * batch["input_mask"]
        batch["labels"] = torch.randint(1, num_labels, batch["labels"].shape, generator=g) * batch["input_mask"]

    # ---- reference forward/backward -------------------------------------------------------------
    model.zero_grad()
    r = _ref_compose(model, batch, regions, variant)
    if regions == 49 and variant == "cl":
        # the model's own forward (hard-coded 49 regions) must equal the composition of its sub-modules
        vis4 = batch["visual_embeds_att"]
        if vis4.dim() == 3:
            vis4 = vis4.permute(0, 2, 1).contiguous().view(-1, 2048, 7, 7)
        model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["added_attention_mask"],
              batch["visual_embeds_mean"], vis4, None)
        _check(name + " model.forward vs composition", model.crf.emissions, r["logits"], 1e-6)
    loss = O.token_ce_loss(r["logits"], batch["labels"], batch["input_mask"])
    loss.backward()
    ref_grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}

    # ---- oracle on the same weights ---------------------------------------------------------------
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()
         if torch.is_floating_point(v)}
    seq, cross, pooled = O.mner_trunk(P, ocfg, batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                                      batch["added_attention_mask"], batch["visual_embeds_att"], layer_num1,
                                      regions, training=False)
    if variant == "cl":
        logits = O.gated_head_cl(P, seq, cross)
    else:
        logits, crs = O.gated_head_gate_cl(P, seq, cross)
        _check(name + " crs", crs, r["crs"])
    oloss = O.token_ce_loss(logits, batch["labels"], batch["input_mask"])
    oloss.backward()
    errs = {"seq": _check(name + " seq", seq, r["seq"]), "cross": _check(name + " cross", cross, r["cross"]),
            "logits": _check(name + " logits", logits, r["logits"]),
            "pooled": _check(name + " pooled", pooled, r["pooled"]),
            "loss": _check(name + " loss", oloss, loss)}
    gerr = 0.0
    for k, gref in ref_grads.items():
        if P[k].grad is None:
            assert gref.abs().max().item() == 0.0, k
            continue
        e = (P[k].grad - gref).abs().max().item()
        rel = e / (gref.abs().max().item() + 1e-12)
        if rel > 1e-4:
            print("  grad mismatch", k, e, gref.abs().max().item())
        gerr = max(gerr, rel)
    assert gerr <= 1e-4, "grad mismatch %.3e" % gerr
    errs["grad_rel"] = gerr

    # ---- write fixture ----------------------------------------------------------------------------
    out = {
        "meta_cfg": np.array([ref_cfg.vocab_size, ref_cfg.hidden_size, ref_cfg.num_hidden_layers,
                              ref_cfg.num_attention_heads, ref_cfg.intermediate_size,
                              ref_cfg.max_position_embeddings, ref_cfg.type_vocab_size, layer_num1,
                              num_labels, regions], dtype=np.int64),
        "meta_variant": np.array(variant),
        "input_ids": batch["input_ids"].numpy(), "segment_ids": batch["segment_ids"].numpy(),
        "input_mask": batch["input_mask"].numpy(),
        "added_attention_mask": batch["added_attention_mask"].numpy(),
        "labels": batch["labels"].numpy(),
        "vis_seed": np.array([batch_kw.get("seed", synth.REFERENCE_SEED)], dtype=np.int64),
        "vis_layout": np.array(batch_kw.get("layout", "BRC")),
        "vis_sample": _sample(batch["visual_embeds_att"]),
        "logits": r["logits"].detach().numpy(),
        "loss": np.array([loss.item()], dtype=np.float64),
        "pooled": r["pooled"].detach().numpy(),
    }
    if full:
        out["seq"] = r["seq"].detach().numpy()
        out["cross"] = r["cross"].detach().numpy()
        out["vis"] = r["vis"].detach().numpy()
        out["gate"] = r["gate"].detach().numpy()
    else:
        out["seq_head"] = r["seq"][:, :4].detach().numpy()
        out["cross_head"] = r["cross"][:, :4].detach().numpy()
    names = sorted(ref_grads)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([ref_grads[k].norm().item() for k in names], dtype=np.float64)
    out["grad_samples"] = np.stack([np.resize(_sample(ref_grads[k], 16), 16) for k in names])
    if full:
        for k in names:
            if ref_grads[k].numel() <= 4096:
                out["grad/" + k] = ref_grads[k].numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB  oracle-vs-reference: %s" % (name, os.path.getsize(path) / 1024.0,
                                                       {k: "%.1e" % v for k, v in errs.items()}))


def make_blocks_case(CM, name):
    """Cross_Modal_Interaction_Module's own BertModel / BertCrossEncoder / cls_layer_both gate on a tiny config."""
    ref_cfg, ocfg = _cfg_pair(CM, vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                              intermediate_size=256, max_position_embeddings=64)

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.bert = CM.BertModel(ref_cfg)
            self.txt2img_attention = CM.BertCrossEncoder(ref_cfg, 2)
            self.cls_layer = CM.cls_layer_both(128, 128)
            self.aux_head = nn.Linear(128, 1)

    m = Holder()
    synth.fill_module_(m)
    m.eval()
    batch = synth.synthetic_batch(2, 32, 49, vocab_size=512, seed=11)
    g = torch.Generator().manual_seed(5)
    s2 = torch.empty(2, 49, 128).normal_(0, 1, generator=g)
    tok = torch.empty(2, 32, 128).normal_(0, 1, generator=g)
    with torch.no_grad():
        layers, pooled = m.bert(batch["input_ids"], batch["segment_ids"], batch["input_mask"])
        img_mask = (1.0 - batch["added_attention_mask"][:, :49][:, None, None, :].float()) * -10000.0
        cross_all = m.txt2img_attention(layers[-1], s2, img_mask)
        feat = m.cls_layer(cross_all[-1][:, 0], tok[:, 0])
        gsig = torch.sigmoid(m.aux_head(feat)).view(2, 1, 1)
        blended = gsig * tok + (1 - gsig) * cross_all[-1]
        P = {k: v for k, v in m.state_dict().items()}
        olayers, opooled = O.bert_model(P, "bert", batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                                        ocfg, all_layers=True)
        ocross = O.cross_encoder(P, "txt2img_attention", olayers[-1], s2, img_mask, ocfg, 2, False)
        oblend = O.scalar_gate_cross_modal(P, ocross[-1], tok)
    errs = {"layers": max(_check("layer%d" % i, a, b) for i, (a, b) in enumerate(zip(olayers, layers))),
            "pooled": _check("pooled", opooled, pooled),
            "cross": max(_check("cross%d" % i, a, b) for i, (a, b) in enumerate(zip(ocross, cross_all))),
            "blend": _check("blend", oblend, blended)}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, input_ids=batch["input_ids"].numpy(), segment_ids=batch["segment_ids"].numpy(),
                        input_mask=batch["input_mask"].numpy(),
                        added_attention_mask=batch["added_attention_mask"].numpy(),
                        s2=s2.numpy(), tok=tok.numpy(),
                        layers=torch.stack(layers).numpy(), pooled=pooled.numpy(),
                        cross=torch.stack(cross_all).numpy(), blended=blended.numpy())
    print("%-28s %8.1f KB  oracle-vs-reference: %s" % (name, os.path.getsize(path) / 1024.0,
                                                       {k: "%.1e" % v for k, v in errs.items()}))


TINY = dict(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
            max_position_embeddings=64)
BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
            intermediate_size=3072, max_position_embeddings=512)


def main():
    torch.set_num_threads(8)
    CM, CL, GCL = _import_reference()
    make_blocks_case(CM, "tiny_blocks")
    make_case(CL, GCL, "tiny_cl_r49", TINY, dict(batch=2, seq_len=32, regions=49, layout="BCHW", seed=1),
              layer_num1=1, num_labels=13, full=True)
    make_case(CL, GCL, "tiny_cl_masks", TINY, dict(batch=4, seq_len=32, regions=49, layout="BRC", seed=2),
              layer_num1=2, num_labels=13, full=True, special_masks=True)
    make_case(CL, GCL, "tiny_gatecl_s128", TINY | dict(max_position_embeddings=128),
              dict(batch=2, seq_len=128, regions=49, layout="BRC", seed=3),
              layer_num1=1, num_labels=13, variant="gate_cl", full=True)
    make_case(CL, GCL, "base_cl_s128_r49", BASE, dict(batch=2, seq_len=128, regions=49, layout="BCHW", seed=4),
              layer_num1=1, num_labels=13)
    make_case(CL, GCL, "base_cl_s64_r36", BASE, dict(batch=2, seq_len=64, regions=36, layout="BRC", seed=5),
              layer_num1=1, num_labels=13)


if __name__ == "__main__":
    main()
