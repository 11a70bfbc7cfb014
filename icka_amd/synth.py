"""Seeded synthetic weights and Twitter-2015-shaped batches.

Weights are generated **by state_dict key** (seed = crc32(key)) so that the reference modules (dev container
only), the CPU oracle and the HIP product all see bit-identical fp32 parameters without any checkpoint
travelling to the GPU box.  Everything is generated on the CPU generator and then moved, so values do not
depend on the device.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Mapping, Optional, Tuple

import torch

REFERENCE_SEED = 19260817  # My_cross_attention.py:577-580 (reference default --seed)


def seeded_tensor(key: str, shape: Tuple[int, ...], std: float = 0.02) -> torch.Tensor:
    """normal(0, std) from a per-key generator; LayerNorm scales are 1 + normal(0, std) so they are exercised."""
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    t = torch.empty(shape, dtype=torch.float32).normal_(0.0, std, generator=g)
    if key.endswith("LayerNorm.weight") or key.endswith("proj_norm.weight"):
        t += 1.0
    return t


def seeded_state_dict(shapes: Mapping[str, Tuple[int, ...]], std: float = 0.02) -> Dict[str, torch.Tensor]:
    return {k: seeded_tensor(k, tuple(shapes[k]), std) for k in sorted(shapes)}


def fill_module_(module: torch.nn.Module, std: float = 0.02, only: Optional[Iterable[str]] = None) -> None:
    """In-place seeded fill of every parameter of ``module`` (keyed by its state_dict names)."""
    sd = module.state_dict()
    names = set(only) if only is not None else None
    with torch.no_grad():
        for k, v in sd.items():
            if names is not None and k not in names:
                continue
            if not torch.is_floating_point(v):
                continue
            v.copy_(seeded_tensor(k, tuple(v.shape), std).to(v.dtype))


def seeded_resnet_tensor(key: str, shape: Tuple[int, ...]) -> torch.Tensor:
    """By-key seeded values for the ResNet image encoder (state_dict names of resnet/resnet.py): convolution weights
    normal(0, sqrt(2/n)) as the reference initialises them (:113-119), BatchNorm affine / running statistics drawn so
    that eval-mode BatchNorm is a non-trivial, well-conditioned affine map (running_var in [0.5, 1.5]; the last
    BatchNorm of a block scaled down so the residual stream of 50 blocks stays in range)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode("utf-8")))
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return 0.5 + torch.rand(shape, generator=g)
    if leaf == "running_mean":
        return torch.empty(shape).normal_(0.0, 0.1, generator=g)
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if len(shape) == 4:   # convolution [out, in, kh, kw]
        n = shape[2] * shape[3] * shape[0]
        return torch.empty(shape).normal_(0.0, (2.0 / n) ** 0.5, generator=g)
    if len(shape) == 2:   # fc
        return torch.empty(shape).normal_(0.0, 0.01, generator=g)
    if leaf == "weight":  # BatchNorm scale
        last = ".bn3." in key or key.endswith("bn3.weight")
        return (0.2 if last else 0.8) + 0.4 * torch.rand(shape, generator=g)
    return torch.empty(shape).normal_(0.0, 0.1, generator=g)   # BatchNorm / fc bias


def fill_resnet_(module: torch.nn.Module) -> None:
    with torch.no_grad():
        for k, v in module.state_dict().items():
            v.copy_(seeded_resnet_tensor(k, tuple(v.shape)).to(v.dtype))


def synthetic_batch(batch: int, seq_len: int, regions: int, num_labels: int = 13, vocab_size: int = 30522,
                    seed: int = REFERENCE_SEED, layout: str = "BRC", ragged: bool = True,
                    min_len: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """One (sentence, image)-pair batch in the reference's tensor conventions.

    input_ids [B,S] int64 ~ U[1,vocab) with pad id 0 past the length; lengths ~ U[S/4,S] (``ragged``);
    segment_ids zeros (My_cross_attention.py:362); input_mask = pos < len; added_attention_mask =
    ones[R] || input_mask (My_cross_attention.py:373); region features ~ N(0,1) either as ``[B,R,2048]``
    (layout "BRC", BASELINE synthetic layout) or as ``[B,2048,7,7]`` (layout "BCHW", the reference's
    myResnet output, only for R=49); labels ~ U[1,C) on valid tokens, 0 elsewhere.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    lo = max(1, seq_len // 4) if min_len is None else min_len
    if ragged:
        lengths = torch.randint(lo, seq_len + 1, (batch,), generator=g)
    else:
        lengths = torch.full((batch,), seq_len, dtype=torch.long)
    pos = torch.arange(seq_len)[None, :]
    input_mask = (pos < lengths[:, None]).long()
    ids = torch.randint(1, vocab_size, (batch, seq_len), generator=g) * input_mask
    segment_ids = torch.zeros_like(ids)
    labels = torch.randint(1, num_labels, (batch, seq_len), generator=g) * input_mask
    if layout == "BRC":
        vis = torch.empty(batch, regions, 2048).normal_(0.0, 1.0, generator=g)
    elif layout == "BCHW":
        if regions != 49:
            raise ValueError("BCHW layout is the reference's 7x7 grid: regions must be 49")
        vis = torch.empty(batch, 2048, 7, 7).normal_(0.0, 1.0, generator=g)
    else:
        raise ValueError(layout)
    vis_mean = torch.empty(batch, 2048).normal_(0.0, 1.0, generator=g)
    added = torch.cat([torch.ones(batch, regions, dtype=torch.long), input_mask], dim=1)
    return {
        "input_ids": ids, "segment_ids": segment_ids, "input_mask": input_mask,
        "added_attention_mask": added, "visual_embeds_mean": vis_mean, "visual_embeds_att": vis,
        "labels": labels, "lengths": lengths,
    }


def synthetic_prompt_batch(batch: int, seq_len: int = 128, vocab_size: int = 30522, roberta_vocab: int = 50265,
                           prompt_tokens: int = 17, total_len: int = 170, num_labels: int = 13,
                           seed: int = REFERENCE_SEED, mask_positions=(3, 11), mask_id: Optional[int] = None,
                           ) -> Dict[str, torch.Tensor]:
    """Inputs of the reference's current model (Cross_Modal_Interaction_Module.py:941-943) in the conventions of
    My_cross_attention.py:291-420: the text-encoder view (``ori_*``, ``seq_len`` tokens, as ``synthetic_batch``) and the
    prompt-encoder view ``input_ids`` = ``prompt_tokens`` prompt-text ids (``<mask>`` at ``mask_positions``) followed by
    the sentence's ids, zero-padded to ``total_len`` (= max_seq_length + prompt words + 30, :305); ``offsets`` =
    ``prompt_tokens`` for every sample (:395, asserted equal per batch at :802); CLIP feature [B,1,512]."""
    b = synthetic_batch(batch, seq_len, 49, num_labels=num_labels, vocab_size=vocab_size, seed=seed, layout="BCHW")
    g = torch.Generator(device="cpu")
    g.manual_seed(seed + 1)
    mask_id = roberta_vocab - 1 if mask_id is None else mask_id
    prompt = torch.randint(3, roberta_vocab - 1, (prompt_tokens,), generator=g)
    for p in mask_positions:
        prompt[p] = mask_id
    sent = torch.randint(3, roberta_vocab - 1, (batch, seq_len), generator=g) * b["input_mask"] \
        + (1 - b["input_mask"])                                        # RoBERTa pad id 1 past the length
    ids = torch.ones(batch, total_len, dtype=torch.long)
    ids[:, :prompt_tokens] = prompt[None]
    ids[:, prompt_tokens:prompt_tokens + seq_len] = sent
    mask = torch.zeros(batch, total_len, dtype=torch.long)
    mask[:, :prompt_tokens] = 1
    mask[:, prompt_tokens:prompt_tokens + seq_len] = b["input_mask"]
    out = {
        "input_ids": ids, "segment_ids": torch.zeros_like(ids), "input_mask": mask,
        "ori_input_ids": b["input_ids"], "ori_input_mask": b["input_mask"], "ori_segment_ids": b["segment_ids"],
        "added_attention_mask": b["added_attention_mask"],
        "clip_features": torch.empty(batch, 1, 512).normal_(0.0, 1.0, generator=g),
        "visual_embeds_mean": b["visual_embeds_mean"], "visual_embeds_att": b["visual_embeds_att"],
        "offsets": torch.full((batch,), prompt_tokens, dtype=torch.long), "output_mask": b["input_mask"],
        "labels": b["labels"],
    }
    return out
