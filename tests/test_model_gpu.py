"""GPU: the drop-in modules (HIP kernels through the C-ABI) against the golden fixtures made by the reference and
against the CPU oracle on the same seeded inputs.  Tolerances: logits 2e-2 abs (BASELINE.json north_star, bf16)."""
import numpy as np
import pytest
import torch

from icka_amd import synth
from golden_util import GOLDEN_DIR, load_case

pytestmark = pytest.mark.gpu

LOGIT_TOL = 2e-2   # bf16 tolerance stated by BASELINE.json:north_star
# per fixture: (relative error of each parameter-gradient L2 norm, relative L2 error of the sampled gradient tensors)
# = 2x the values measured on MI355X (the test prints the measured values and the worst key)
C4_GEOMETRY_GRAD_BAR = 1.6e-2   # measured 7.8e-3 (3.0e-2 before the tiled-attention delta fix)
ALIGN_GRAD_BARS = (1.2e-2, 2.3e-2)   # measured 5.9e-3 / 1.15e-2   # single-query alignment encoders: (input gradients, parameter gradients)
# measured: tiny_* 5.4e-3 .. 8.5e-3 (norms and tensors); base_* norms <= 1.3e-2 (the base fixtures hold norms only)
GRAD_BARS = {"tiny_cl_r49": (1.7e-2, 1.7e-2), "tiny_cl_masks": (1.6e-2, 1.6e-2), "tiny_gatecl_s128": (1.1e-2, 1.6e-2),
             "base_cl_s64_r36": (2.6e-2, 2.6e-2), "base_cl_s128_r49": (2.6e-2, 2.6e-2)}


def _build(cfg, regions=49, variant="cl", max_seq_length=128):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    c = BertConfig(cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                   num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                   max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"])
    m = MTCCMBertForMMTokenClassificationCRF(c, layer_num1=cfg["layer_num1"], num_labels=cfg["num_labels"],
                                             regions=regions, variant=variant, max_seq_length=max_seq_length)
    synth.fill_module_(m)
    return m.cuda()


def _run(model, batch, labels=True):
    g = {k: v.cuda() for k, v in batch.items()}
    return model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"],
                 g["visual_embeds_mean"], g["visual_embeds_att"], labels=g["labels"] if labels else None)


@pytest.mark.parametrize("name", ["tiny_cl_r49", "tiny_cl_masks", "tiny_gatecl_s128", "base_cl_s64_r36",
                                  "base_cl_s128_r49"])
def test_logits_loss_and_grads_match_reference_fixture(name):
    case = load_case(name)
    exp = case["expected"]
    model = _build(case["cfg"], case["cfg"]["regions"], variant=case["variant"],
                   max_seq_length=case["batch"]["input_ids"].shape[1]).eval()
    logits = _run(model, case["batch"], labels=False)
    assert logits.dtype == torch.float32 and tuple(logits.shape) == exp["logits"].shape
    err = np.abs(logits.detach().cpu().numpy() - exp["logits"]).max()
    assert err < LOGIT_TOL, "logits max abs err %.3e" % err
    print("\n[%s] logits max abs err %.3e (tol %.0e)" % (name, err, LOGIT_TOL))
    model.zero_grad()
    loss = _run(model, case["batch"], labels=True)
    assert abs(loss.item() - float(exp["loss"][0])) < LOGIT_TOL
    loss.backward()
    params = dict(model.named_parameters())
    gmax = float(exp["grad_norms"].max())   # e.g. key.bias has an exactly-zero true gradient: absolute floor
    worst_n, worst_t, key_n, key_t = 0.0, 0.0, "", ""
    for n, gn in zip([str(x) for x in exp["grad_names"]], exp["grad_norms"]):
        if n not in params or gn == 0.0:
            continue
        g = params[n].grad
        assert g is not None, n
        rel = abs(g.float().norm().item() - gn) / (gn + 1e-4 * gmax)
        if rel > worst_n:
            worst_n, key_n = rel, n
        key = "grad/" + n
        if key in exp:
            ref = torch.from_numpy(exp[key])
            e = ((g.float().cpu() - ref).norm() / (ref.norm() + 1e-4 * gmax)).item()
            if e > worst_t:
                worst_t, key_t = e, n
    bar_n, bar_t = GRAD_BARS[name]
    print("[%s] worst gradient-norm error %.3e at %s (bar %.1e); worst gradient-tensor rel-L2 %.3e at %s (bar %.1e)"
          % (name, worst_n, key_n, bar_n, worst_t, key_t, bar_t))
    assert worst_n < bar_n, (key_n, worst_n)
    assert worst_t < bar_t, (key_t, worst_t)
    assert model.bert.embeddings.word_embeddings.weight.grad[0].abs().max().item() == 0.0   # padding_idx row


def test_blocks_against_fixture():
    """BertModel (all layers + pooler) and a 2-layer BertCrossEncoder, reference block API."""
    from icka_amd.config import BertConfig
    from icka_amd.modeling import BertCrossEncoder, BertModel
    z = np.load(GOLDEN_DIR + "/tiny_blocks.npz")
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.bert = BertModel(cfg)
            self.txt2img_attention = BertCrossEncoder(cfg, 2)

    m = Holder()
    synth.fill_module_(m)
    m = m.cuda().eval()
    ids, seg, msk = (torch.from_numpy(z[k]).cuda() for k in ("input_ids", "segment_ids", "input_mask"))
    layers, pooled = m.bert(ids, seg, msk)
    assert len(layers) == 2
    for i, l in enumerate(layers):
        assert np.abs(l.detach().float().cpu().numpy() - z["layers"][i]).max() < 3e-2
    assert np.abs(pooled.detach().float().cpu().numpy() - z["pooled"]).max() < 2e-2
    img = (1.0 - torch.from_numpy(z["added_attention_mask"])[:, :49].float())[:, None, None, :] * -10000.0
    cross = m.txt2img_attention(torch.from_numpy(z["layers"][-1]).cuda(), torch.from_numpy(z["s2"]).cuda(), img.cuda())
    assert len(cross) == 2
    for i, c in enumerate(cross):
        assert np.abs(c.detach().float().cpu().numpy() - z["cross"][i]).max() < 3e-2
    last_only = m.bert(ids, seg, msk, output_all_encoded_layers=False)[0]
    assert torch.equal(last_only, layers[-1])


def test_scalar_gate_fusion_against_fixture():
    """Cross_Modal form of the gate (cls_layer_both + aux_head, :1029-1036) with token_embedding as an input."""
    from icka_amd.modeling import cls_layer_both, scalar_gate_fusion
    z = np.load(GOLDEN_DIR + "/tiny_blocks.npz")

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cls_layer = cls_layer_both(128, 128)
            self.aux_head = torch.nn.Linear(128, 1)

    m = Holder()
    synth.fill_module_(m)          # same key-seeded values (and the same proj_norm/LayerNorm aliasing) as the fixture
    m = m.cuda()
    cross = torch.from_numpy(z["cross"][-1]).cuda().requires_grad_(True)
    tok = torch.from_numpy(z["tok"]).cuda().requires_grad_(True)
    out = scalar_gate_fusion(m, cross, tok)
    assert np.abs(out.detach().float().cpu().numpy() - z["blended"]).max() < 3e-2
    out.float().sum().backward()
    # reference math on the same bf16-rounded inputs (fp32 torch) for the gradients
    c32 = cross.detach().to(torch.bfloat16).float().requires_grad_(True)
    t32 = tok.detach().to(torch.bfloat16).float().requires_grad_(True)
    P = {k: v.detach().float() for k, v in m.state_dict().items()}
    feat = torch.nn.functional.layer_norm(c32[:, 0] + t32[:, 0], (128,), P["cls_layer.proj_norm.weight"],
                                          P["cls_layer.proj_norm.bias"], 1e-5)
    g = torch.sigmoid((feat @ P["cls_layer.proj.weight"].t() + P["cls_layer.proj.bias"]) @ P["aux_head.weight"].t()
                      + P["aux_head.bias"]).view(-1, 1, 1)
    (g * t32 + (1 - g) * c32).sum().backward()
    for mine, ref in ((cross.grad, c32.grad), (tok.grad, t32.grad)):
        assert ((mine.float() - ref).norm() / ref.norm()).item() < 3e-2
    assert m.aux_head.weight.grad is not None and torch.isfinite(m.aux_head.weight.grad).all()


def test_train_mode_dropout_statistics_and_determinism():
    """Dropout cannot match the CPU RNG stream: check it is active, seeded and reproducible, and that the backward
    uses the forward's masks (finite-difference-free check: two identical seeded runs give identical grads)."""
    case = load_case("tiny_cl_r49")
    model = _build(case["cfg"]).train()
    outs = []
    for _ in range(2):
        model._icka_arena.set_seed(1234) if hasattr(model, "_icka_arena") else None
        model.zero_grad()
        loss = _run(model, case["batch"])
        if not outs:
            model._icka_arena.set_seed(1234)   # arena exists after the first forward: rerun from a known seed
            model.zero_grad()
            loss = _run(model, case["batch"])
        loss.backward()
        outs.append((loss.item(), model.classifier.weight.grad.clone(), model.vismap2text.weight.grad.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    model.eval()
    eval_loss = _run(model, case["batch"]).item()
    assert abs(eval_loss - outs[0][0]) > 1e-6       # dropout changed the result
    assert abs(eval_loss - outs[0][0]) < 1.0


def test_attention_keep_bits_option_gives_the_same_step(monkeypatch):
    """ICKA_ATTN_KEEPBITS=1 (icka_amd.ops.ATTN_KEEPBITS): the attention forward leaves its dropout decisions as bits and the
    backward reads them instead of hashing again -- same masks, so a seeded train-mode step is bitwise the default one."""
    from icka_amd import ops
    case = load_case("tiny_cl_r49")
    model = _build(case["cfg"]).train()
    _run(model, case["batch"]).backward()                  # builds the arena
    outs = []
    for keep in ("0", "1"):
        monkeypatch.setattr(ops, "ATTN_KEEPBITS", keep)
        model._icka_arena.set_seed(4321)
        model.zero_grad()
        loss = _run(model, case["batch"])
        loss.backward()
        outs.append((loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert outs[0][0] == outs[1][0]
    assert outs[0][1].keys() == outs[1][1].keys()
    for n, g in outs[0][1].items():
        if "word_embeddings" in n:                          # (f32 atomics: order-dependent in the last bits)
            assert torch.allclose(g, outs[1][1][n], rtol=1e-4, atol=1e-7), n
        else:
            assert torch.equal(g, outs[1][1][n]), n


def test_grad_accumulation_and_zero_grad():
    case = load_case("tiny_cl_r49")
    model = _build(case["cfg"]).eval()
    _run(model, case["batch"]).backward()
    g1 = model.classifier.weight.grad.clone()
    _run(model, case["batch"]).backward()                  # second micro-batch accumulates (My_cross_attention.py:831)
    assert torch.allclose(model.classifier.weight.grad, 2 * g1, rtol=1e-3, atol=1e-6)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    opt.zero_grad()                                        # set_to_none=True: next backward starts fresh
    _run(model, case["batch"]).backward()
    assert torch.allclose(model.classifier.weight.grad, g1, rtol=1e-3, atol=1e-6)
    before = model.classifier.weight.detach().clone()
    opt.step()                                             # parameters are arena views; the shadow must refresh
    assert not torch.equal(before, model.classifier.weight.detach())
    l2 = _run(model, case["batch"]).item()
    assert np.isfinite(l2)


def test_cpu_inputs_are_refused():
    case = load_case("tiny_cl_r49")
    model = _build(case["cfg"]).eval()
    b = case["batch"]
    with pytest.raises(TypeError):
        model(b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"], None, b["visual_embeds_att"])


@pytest.mark.parametrize("layout,regions", [("BRC", 50), ("BCHW", 49)])
def test_bert_large_geometry_against_live_oracle(layout, regions):
    """BASELINE config c4 geometry (bert-large: H 1024, 16 heads, I 4096, seq 256, 50 regions) at 2 layers / batch 2:
    exercises the tiled (Sq, Skv > 128) attention kernels, NCH=2 LayerNorm rows of 1024 and the 50-region
    cross-attention; compared with the CPU oracle run on the same seeded weights and inputs."""
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    from oracle import mner_oracle as O
    S, B = 256, 2
    cfg = BertConfig(2048, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                     max_position_embeddings=512)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=regions, max_seq_length=S)
    synth.fill_module_(model)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    b = synth.synthetic_batch(B, S, regions, vocab_size=2048, seed=11, layout=layout)
    logits = _run(model, b, labels=False)
    loss = _run(model, b, labels=True)
    loss.backward()
    ocfg = O.OracleConfig(vocab_size=2048, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16,
                          intermediate_size=4096, max_position_embeddings=512)
    ref = O.mner_logits(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                        b["visual_embeds_att"], 1, regions)
    rloss = O.token_ce_loss(ref, b["labels"], b["input_mask"])
    rloss.backward()
    err = (logits.float().cpu() - ref.detach()).abs().max().item()
    assert err < LOGIT_TOL, "logits max abs err %.3e" % err
    assert abs(loss.item() - rloss.item()) < LOGIT_TOL
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst = 0.0
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        gr = P[k].grad
        worst = max(worst, ((p.grad.float().cpu() - gr).norm() / (gr.norm() + 1e-4 * gmax)).item())
    print("\n[c4 geometry %s R=%d] logits max abs err %.3e, worst grad rel err %.3e" % (layout, regions, err, worst))
    assert worst < C4_GEOMETRY_GRAD_BAR, "worst relative gradient error %.3e" % worst


@pytest.mark.parametrize("hidden,heads,precision", [(128, 4, "bf16"), (256, 2, "bf16"), (192, 4, "mixed16"), (128, 4, "fp32")])
def test_head_sizes_other_than_64_against_live_oracle(hidden, heads, precision):
    """VERDICT r03 missing #4: the reference accepts any hidden % heads == 0 (Cross_Modal_Interaction_Module.py:459-462).  Head
    sizes 32 / 128 / 48 run the 16-bit modes with the attention core on the f32-input MFMA kernels (ops._attn_generic_fwd /
    _bwd): self- and co-attention, ragged masks, train-mode determinism; logits and every gradient against the CPU oracle."""
    import icka_amd
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    from oracle import mner_oracle as O
    S, B, R = 32, 4, 36
    cfg = BertConfig(512, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=2 * hidden,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=R, max_seq_length=S)
    synth.fill_module_(model)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = icka_amd.set_precision(model.cuda().eval(), precision)
    b = synth.synthetic_batch(B, S, R, vocab_size=512, seed=11)
    logits = _run(model, b, labels=False)
    model.zero_grad()
    loss = _run(model, b, labels=True)
    loss.backward()
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads,
                          intermediate_size=2 * hidden, max_position_embeddings=64)
    ref = O.mner_logits(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                        b["visual_embeds_att"], 1, R)
    rloss = O.token_ce_loss(ref, b["labels"], b["input_mask"])
    rloss.backward()
    tol = 1e-4 if precision == "fp32" else LOGIT_TOL
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
    print("\n[head size %d (H%d / %d heads), %s] logits max abs err %.3e (tol %.0e), loss %.5f (oracle %.5f), worst gradient "
          "rel-L2 %.3e at %s" % (hidden // heads, hidden, heads, precision, err, tol, loss.item(), rloss.item(), worst, wkey))
    assert err < tol and abs(loss.item() - rloss.item()) < tol
    assert worst < (1e-3 if precision == "fp32" else 3e-2), (worst, wkey)
    if precision != "fp32":     # train mode: the dropout of the generic core follows the same seeds -> deterministic steps
        model.train()
        outs = []
        for _ in range(2):
            model._icka_arena.set_seed(77)
            model.zero_grad()
            l2 = _run(model, b, labels=True)
            l2.backward()
            outs.append((l2.item(), model.classifier.weight.grad.clone()))
        assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
        assert outs[0][0] != loss.item()


def test_fp8_cross_attention_reference_diff_report():
    """BASELINE config c5 (bert-base, seq 128, 36 regions, fp8 QK^T/PV in the cross-attention): logits of the fp8
    variant against the fp32 CPU oracle and against the bf16 path, on the by-key seeded weights.  Reports the
    differences; the fp8 variant must stay inside the bf16 tolerance budget of north_star on this workload."""
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    from oracle import mner_oracle as O
    B, S, R, L = 4, 128, 36, 2
    cfg = BertConfig(4096, hidden_size=768, num_hidden_layers=L, num_attention_heads=12, intermediate_size=3072,
                     max_position_embeddings=512)
    outs = {}
    for fp8 in (False, True):
        model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=R,
                                                     cross_attention_fp8=fp8)
        synth.fill_module_(model)
        P = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model = model.cuda().eval()
        b = synth.synthetic_batch(B, S, R, vocab_size=4096, seed=5)
        outs[fp8] = _run(model, b, labels=False).float().cpu()
        if fp8:   # the backward of the fp8 forward runs (bf16 recomputation) and gives finite gradients
            _run(model, b, labels=True).backward()
            assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    ocfg = O.OracleConfig(vocab_size=4096, hidden_size=768, num_hidden_layers=L, num_attention_heads=12,
                          intermediate_size=3072, max_position_embeddings=512)
    with torch.no_grad():
        ref = O.mner_logits(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                            b["visual_embeds_att"], 1, R)
    e16 = (outs[False] - ref).abs().max().item()
    e8 = (outs[True] - ref).abs().max().item()
    d = (outs[True] - outs[False]).abs().max().item()
    print("\n[c5 fp8 cross-attention] max |dlogits| vs fp32 oracle: bf16 path %.3e, fp8 path %.3e; fp8 vs bf16 %.3e"
          % (e16, e8, d))
    assert e16 < LOGIT_TOL and e8 < 2.5 * LOGIT_TOL


def test_alignment_cross_encoder_single_query_token():
    """the alignment cross-encoders (`cls_layer_Y`, Cross_Modal_Interaction_Module.py
    :901, :981-989) call BertCrossEncoder with ONE query token per sample (the mapped CLIP feature, S_q = 1) attending
    over the S text tokens under the ragged text mask.  Same kernels at a degenerate shape (M = B rows), forward and
    backward, against the CPU oracle's cross_encoder."""
    from icka_amd.config import BertConfig
    from icka_amd.modeling import BertCrossEncoder
    from oracle import mner_oracle as O
    B, S, H = 6, 128, 256
    cfg = BertConfig(512, hidden_size=H, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
                     max_position_embeddings=128)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cls_layer_Y = torch.nn.ModuleList([BertCrossEncoder(cfg, 1), BertCrossEncoder(cfg, 1)])

    m = Holder()
    synth.fill_module_(m)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(7)
    clip = torch.randn(B, 1, H, generator=g)
    text = torch.randn(B, S, H, generator=g)
    lens = torch.randint(S // 4, S + 1, (B,), generator=g)
    mask01 = (torch.arange(S)[None, :] < lens[:, None]).long()
    ext = ((1.0 - mask01.float()) * -10000.0)[:, None, None, :]
    c_gpu = clip.cuda().requires_grad_(True)
    t_gpu = text.cuda().requires_grad_(True)
    x = c_gpu
    for enc in m.cls_layer_Y:
        x = enc(x, t_gpu, ext.cuda())[-1]
    assert tuple(x.shape) == (B, 1, H)
    x.float().sum().backward()
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=H, num_hidden_layers=1, num_attention_heads=4,
                          intermediate_size=512, max_position_embeddings=128)
    c_ref = clip.clone().requires_grad_(True)
    t_ref = text.clone().requires_grad_(True)
    r = c_ref
    for i in range(2):
        r = O.cross_encoder(P, "cls_layer_Y.%d" % i, r, t_ref, ext, ocfg, 1, False)[-1]
    r.sum().backward()
    err = (x.detach().float().cpu() - r.detach()).abs().max().item()
    assert err < 3e-2, err
    win = max(((mine.float().cpu() - ref).norm() / ref.norm()).item()
              for mine, ref in ((c_gpu.grad, c_ref.grad), (t_gpu.grad, t_ref.grad)))
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    wpar, wkey = 0.0, ""
    for k, p in m.named_parameters():
        if P[k].grad is not None:
            rel = ((p.grad.float().cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-4 * gmax)).item()
            if rel > wpar:
                wpar, wkey = rel, k
    print("\n[S_q=1 alignment encoders] max abs err %.3e; input-gradient rel-L2 %.3e; worst parameter gradient %.3e at %s"
          % (err, win, wpar, wkey))
    assert win < ALIGN_GRAD_BARS[0] and wpar < ALIGN_GRAD_BARS[1], (win, wpar, wkey)
