"""GPU: ragged and extreme shapes of the path against the live CPU oracle (same seeded weights and inputs): batch 1, sequences
of 1 / 2 / 17 / 129 tokens and the 512 positions the embedding table allows, 1 / 5 / 37 / 256 regions, a sentence with nothing but
its first token, fully padded tails and images whose regions are all masked (the reference builds these masks at
My_cross_attention.py:362-373 and adds -10000 per masked key, Cross_Modal_Interaction_Module.py:969-975: a fully masked row is a
uniform softmax there, and must be here).  Logits, loss and every parameter gradient; fp32 mode at 1e-4 / bf16 at north_star's
2e-2.  None of these shapes is a multiple of a tile of any kernel: the row / key tails of the GEMM, attention, LayerNorm,
embedding and classifier kernels are all exercised through the module API."""
import pytest
import torch

import icka_amd
from icka_amd import synth

pytestmark = pytest.mark.gpu

#          B   S    R   what
SHAPES = [(1, 1, 1, "one token, one region"),
          (1, 2, 5, "two tokens"),
          (3, 17, 5, "odd everything"),
          (2, 129, 37, "one past the whole-head attention tile"),
          (5, 33, 1, "a single region"),
          (1, 512, 36, "every position of the table"),
          (2, 40, 256, "the largest region count the layout kernel takes"),
          (7, 24, 49, "seven pairs")]


def _model(S, R, precision, maxpos):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=maxpos)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=R, max_seq_length=S)
    synth.fill_module_(model)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    return icka_amd.set_precision(model.cuda().eval(), precision), P


def _compare(model, P, b, S, R, precision, maxpos, tag):
    from oracle import mner_oracle as O
    g = {k: v.cuda() for k, v in b.items()}
    args = (g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
            g["visual_embeds_att"])
    logits = model(*args)
    model.zero_grad()
    loss = model(*args, labels=g["labels"])
    loss.backward()
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=maxpos)
    ref = O.mner_logits(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                        b["visual_embeds_att"], 1, R)
    rloss = O.token_ce_loss(ref, b["labels"], b["input_mask"])
    rloss.backward()
    tol = 1e-4 if precision == "fp32" else 2e-2
    assert torch.isfinite(logits).all()
    err = (logits.float().cpu() - ref.detach()).abs().max().item()
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst, wkey = 0.0, ""
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        rel = ((p.grad.float().cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-3 * gmax)).item()
        if rel > worst:
            worst, wkey = rel, k
    print("\n[%s | B%d S%d R%d %s] logits max abs err %.3e (tol %.0e), loss %.5f (oracle %.5f), worst gradient rel-L2 %.3e at %s"
          % (tag, b["input_ids"].shape[0], S, R, precision, err, tol, loss.item(), rloss.item(), worst, wkey))
    assert err < tol and abs(loss.item() - rloss.item()) < tol * max(1.0, abs(rloss.item()))
    assert worst < (1e-3 if precision == "fp32" else 4e-2), (worst, wkey)
    for v in P.values():
        v.grad = None


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("B,S,R,what", SHAPES, ids=["B%dS%dR%d" % s[:3] for s in SHAPES])
def test_ragged_and_extreme_shapes_against_live_oracle(B, S, R, what, precision):
    maxpos = max(64, S)
    model, P = _model(S, R, precision, maxpos)
    b = synth.synthetic_batch(B, S, R, vocab_size=512, seed=31 + S, min_len=1)
    _compare(model, P, b, S, R, precision, maxpos, what)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_degenerate_masks_against_live_oracle(precision):
    """Row 0: only the first token is real.  Row 1: the image is fully masked (every region key carries -10000: the text -> image
    attention is a uniform average over the regions in the reference, and the image -> text rows see ordinary text keys).  Row 2:
    full length.  Row 3: regions 0..R/2 masked."""
    B, S, R = 4, 24, 36
    model, P = _model(S, R, precision, 64)
    b = synth.synthetic_batch(B, S, R, vocab_size=512, seed=77, ragged=False)
    mask = b["input_mask"].clone()
    mask[0, 1:] = 0
    b["input_mask"] = mask
    b["input_ids"] = b["input_ids"] * mask
    b["labels"] = b["labels"] * mask
    added = torch.cat([torch.ones(B, R, dtype=torch.long), mask], dim=1)
    added[1, :R] = 0
    added[3, :R // 2] = 0
    b["added_attention_mask"] = added
    _compare(model, P, b, S, R, precision, 64, "degenerate masks")
