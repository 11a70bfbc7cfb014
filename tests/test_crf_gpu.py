"""GPU: icka_crf_* kernels and the drop-in CRF module against the CPU oracle (oracle/crf_oracle.py) and brute force."""
import pytest
import torch

pytestmark = pytest.mark.gpu

HEAD_CRF_GRAD_BAR = 1.3e-2   # 2x the 6.1e-3 measured on MI355X (printed by the test)

F32 = torch.float32


def _case(B, S, Cn, seed, ragged=True):
    g = torch.Generator().manual_seed(seed)
    e = torch.randn(B, S, Cn, generator=g) * 2.0
    tags = torch.randint(0, Cn, (B, S), generator=g)
    lens = torch.randint(1, S + 1, (B,), generator=g) if ragged else torch.full((B,), S)
    lens[0] = S
    if B > 1:
        lens[1] = 1
    mask = (torch.arange(S)[None, :] < lens[:, None])
    tags = torch.where(mask, tags, torch.zeros_like(tags))   # pad label 0 like the reference's data
    return e, tags, mask, lens


@pytest.mark.parametrize("B,S,Cn", [(4, 5, 3), (32, 128, 13), (8, 128, 15), (3, 256, 9), (2, 7, 64)])
@pytest.mark.parametrize("reduction", ["mean", "token_mean", "sum", "none"])
def test_crf_llh_grad_decode_against_oracle(B, S, Cn, reduction):
    from icka_amd.crf import CRF
    from oracle import crf_oracle as O
    e, tags, mask, lens = _case(B, S, Cn, seed=B * 1000 + S + Cn)
    crf = CRF(Cn, batch_first=True).cuda()
    P = [p.detach().cpu().clone().requires_grad_(True) for p in (crf.start_transitions, crf.end_transitions, crf.transitions)]
    eg = e.cuda().requires_grad_(True)
    out = crf(eg, tags.cuda(), mask=mask.cuda().byte(), reduction=reduction)
    er = e.clone().requires_grad_(True)
    ref = O.crf_reduce(O.crf_llh(er, tags, mask, *P), mask, reduction)
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=2e-5, atol=2e-4), (out, ref)
    w = torch.linspace(0.5, 1.5, B) if reduction == "none" else None
    (-(out * w.cuda()).sum() if w is not None else -out).backward()
    (-(ref * w).sum() if w is not None else -ref).backward()
    # fp32 recursions of S steps through v_exp_f32 / v_log_f32: marginals agree to ~1e-4
    assert (eg.grad.cpu() - er.grad).abs().max().item() < 1e-3 * er.grad.abs().max().item()
    for mine, theirs in zip((crf.start_transitions, crf.end_transitions, crf.transitions), P):
        scale = theirs.grad.abs().max().item() + 1e-6
        assert (mine.grad.cpu() - theirs.grad).abs().max().item() < 1e-3 * scale + 2e-5, (mine.grad, theirs.grad)
    # gradient accumulation (second backward adds), then zero_grad -> overwrite
    out2 = crf(eg, tags.cuda(), mask=mask.cuda().byte(), reduction="sum")
    (-out2).backward()
    ref2 = O.crf_reduce(O.crf_llh(er, tags, mask, *P), mask, "sum")
    (-ref2).backward()
    assert (crf.transitions.grad.cpu() - P[2].grad).abs().max().item() < 1e-3 * (P[2].grad.abs().max().item() + 1e-3)
    # Viterbi
    paths = crf.decode(eg.detach(), mask=mask.cuda().byte())
    assert paths == O.crf_decode(e, mask, *[p.detach() for p in P])
    assert [len(p) for p in paths] == lens.tolist()


@pytest.mark.parametrize("scale,forbid", [(2.0, True), (40.0, False), (60.0, True)])
def test_crf_extreme_scores_against_oracle(scale, forbid):
    """-1e4 "forbidden" transitions (BIO constraints) and emission spreads up to e^+-200: the scaled linear-domain recursion
    either stays in range or hands the sample to the log-domain body -- both must match the oracle."""
    from icka_amd.crf import CRF
    from oracle import crf_oracle as O
    B, S, Cn = 6, 64, 13
    e, tags, mask, lens = _case(B, S, Cn, seed=77)
    e = e * (scale / 2.0)
    torch.manual_seed(int(scale) * 2 + int(forbid))   # (the CRF parameters are drawn at construction)
    crf = CRF(Cn, batch_first=True).cuda()
    with torch.no_grad():
        if forbid:
            g = torch.Generator().manual_seed(5)
            blocked = torch.rand(Cn, Cn, generator=g) < 0.3
            blocked[torch.arange(Cn), torch.arange(Cn)] = False
            crf.transitions[blocked.cuda()] = -1e4
            crf.start_transitions[::3] = -1e4
            # keep the gold paths feasible (finite log-likelihood): route them over allowed transitions only
            ok_next = (~blocked)
            tags[:, 0] = 1
            for b in range(B):
                for t in range(1, S):
                    cand = torch.nonzero(ok_next[tags[b, t - 1]]).flatten()
                    tags[b, t] = cand[(b + t) % len(cand)]
            tags = torch.where(mask, tags, torch.zeros_like(tags))
    # the oracle in float64: at |scores| ~ 7e3 an f32 log-domain recursion (the oracle's own, and the kernels' fallback body)
    # carries ~1e-3 of rounding in the marginals (f32 oracle: marginals of 1.0002)
    P = [p.detach().cpu().double().clone().requires_grad_(True) for p in (crf.start_transitions, crf.end_transitions, crf.transitions)]
    eg = e.cuda().requires_grad_(True)
    out = crf(eg, tags.cuda(), mask=mask.cuda().byte(), reduction="none")
    er = e.double().clone().requires_grad_(True)
    ref = O.crf_llh(er, tags, mask, *P)
    assert torch.isfinite(out).all() and torch.isfinite(ref).all()
    assert torch.allclose(out.detach().cpu().double(), ref.detach(), rtol=3e-5, atol=2e-3 * max(1.0, scale / 10)), (out, ref)
    (-out.sum()).backward()
    (-ref.sum()).backward()
    assert torch.isfinite(eg.grad).all()
    bar = 2e-3 if scale < 40 else 4e-3
    assert (eg.grad.cpu().double() - er.grad).abs().max().item() < bar
    for mine, theirs in zip((crf.start_transitions, crf.end_transitions, crf.transitions), P):
        assert (mine.grad.cpu().double() - theirs.grad).abs().max().item() < bar * (theirs.grad.abs().max().item() + 1.0), (mine.grad, theirs.grad)


def test_crf_decode_matches_brute_force_and_sequence_first_layout():
    from icka_amd.crf import CRF
    from oracle import crf_oracle as O
    e, tags, mask, lens = _case(5, 6, 3, seed=9)
    crf = CRF(3).cuda()        # batch_first=False: [S,B,C] like the package default
    st, en, tr = (p.detach().cpu() for p in (crf.start_transitions, crf.end_transitions, crf.transitions))
    paths = crf.decode(e.transpose(0, 1).cuda(), mask=mask.transpose(0, 1).cuda())
    llh = crf(e.transpose(0, 1).cuda(), tags.transpose(0, 1).cuda(), mask=mask.transpose(0, 1).cuda(), reduction="none")
    ref = O.crf_llh(e, tags, mask, st, en, tr)
    assert torch.allclose(llh.cpu(), ref, rtol=2e-5, atol=2e-4)
    for b in range(5):
        _, best, _ = O.brute_force(e[b], int(lens[b]), st, en, tr)
        assert paths[b] == best


def test_crf_argument_errors():
    from icka_amd.crf import CRF
    crf = CRF(5, batch_first=True).cuda()
    with pytest.raises(ValueError):
        crf(torch.zeros(2, 3, 4, device="cuda"), torch.zeros(2, 3, dtype=torch.long, device="cuda"))
    with pytest.raises(ValueError):
        crf(torch.zeros(2, 3, 5, device="cuda"), torch.zeros(2, 4, dtype=torch.long, device="cuda"))
    with pytest.raises(ValueError):
        crf(torch.zeros(2, 3, 5, device="cuda"), torch.zeros(2, 3, dtype=torch.long, device="cuda"), reduction="avg")
    with pytest.raises(TypeError):
        crf(torch.zeros(2, 3, 5), torch.zeros(2, 3, dtype=torch.long))
    with pytest.raises(ValueError):
        CRF(0)


def test_model_with_native_crf_loss_and_decode():
    """my_bert head end to end with the HIP CRF: loss = -crf(logits, labels, mask, 'mean') (cl_modeling.py:1380) and
    decode when no labels are given (:1386), against the oracle trunk + CRF oracle on the same weights."""
    from icka_amd import synth
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    from oracle import crf_oracle as OC
    from oracle import mner_oracle as O
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, use_crf=True)
    synth.fill_module_(model)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    b = synth.synthetic_batch(4, 32, 49, vocab_size=512, seed=3, layout="BCHW")
    g = {k: v.cuda() for k, v in b.items()}
    args = (g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
            g["visual_embeds_att"])
    loss = model(*args, labels=g["labels"])
    loss.backward()
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=64)
    ref_logits = O.mner_logits(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                               b["visual_embeds_att"], 1, 49)
    crfP = [P["crf.start_transitions"], P["crf.end_transitions"], P["crf.transitions"]]
    rloss = -OC.crf_llh(ref_logits, b["labels"], b["input_mask"].bool(), *crfP).mean()
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 2e-2 * max(1.0, abs(rloss.item()))
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst, wk = 0.0, ""
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        rel = ((p.grad.float().cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-4 * gmax)).item()
        if rel > worst:
            worst, wk = rel, k
    print("\n[gated head + CRF] loss %.4f (oracle %.4f), worst gradient rel-L2 %.3e at %s" % (loss.item(), rloss.item(), worst, wk))
    assert worst < HEAD_CRF_GRAD_BAR, (wk, worst)
    pred = model(*args)
    assert isinstance(pred, list) and [len(p) for p in pred] == b["input_mask"].sum(1).tolist()
    ref_pred = OC.crf_decode(model.logits(*args[:4], args[5]).float().cpu(), b["input_mask"].bool(),
                             *[p.detach() for p in crfP])
    assert pred == ref_pred
