"""Load the committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the
reference itself) and rebuild the full input batch they were computed on."""
from __future__ import annotations

import os
from typing import Dict

import numpy as np
import torch

from icka_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CFG_FIELDS = ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
              "max_position_embeddings", "type_vocab_size", "layer_num1", "num_labels", "regions")


def _sample(t: torch.Tensor, n: int = 64) -> np.ndarray:
    f = t.detach().reshape(-1)
    m = min(n, f.numel())
    idx = (torch.arange(m, dtype=torch.long) * (f.numel() - 1)) // max(m - 1, 1)
    return f[idx].numpy().copy()


def load_case(name: str) -> Dict:
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    cfg = dict(zip(CFG_FIELDS, [int(v) for v in z["meta_cfg"]]))
    b, s = z["input_ids"].shape
    layout = str(z["vis_layout"])
    regen = synth.synthetic_batch(b, s, cfg["regions"], num_labels=cfg["num_labels"], vocab_size=cfg["vocab_size"],
                                  seed=int(z["vis_seed"][0]), layout=layout)
    vis = regen["visual_embeds_att"]
    np.testing.assert_array_equal(_sample(vis), z["vis_sample"])  # same generator stream as when the fixture was made
    batch = {k: torch.from_numpy(z[k]) for k in ("input_ids", "segment_ids", "input_mask", "added_attention_mask",
                                                 "labels")}
    batch["visual_embeds_att"] = vis
    batch["visual_embeds_mean"] = regen["visual_embeds_mean"]
    exp = {k: z[k] for k in z.files}
    return {"cfg": cfg, "variant": str(z["meta_variant"]), "batch": batch, "expected": exp}
