"""GPU: the fp32-exact mode (icka_amd/exact.py, csrc/exact.hip) against the reference-made golden fixtures at the
**1e-3 fp32** tolerance of BASELINE.json:north_star, and its batched f32-MFMA GEMM against fp64 matmul."""
import numpy as np
import pytest
import torch

from icka_amd import synth
from golden_util import GOLDEN_DIR, load_case

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3    # "within 1e-3 fp32" (BASELINE.json north_star)
GRAD_TOL = 2.5e-4  # relative L2 per parameter: 2x the worst measured (1.1e-4, bert-base fixtures; tiny ones 2e-5)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.mark.parametrize("op,M,N,K", [(0, 128, 128, 64), (0, 200, 77, 50), (1, 130, 64, 36), (2, 49, 64, 128),
                                      (3, 36, 768, 2048), (0, 4, 13, 1536), (1, 512, 256, 13), (2, 768, 96, 1000),
                                      (0, 1, 2, 4099)])
def test_xgemm_against_fp64(op, M, N, K):
    from icka_amd import exact as X
    torch.manual_seed(op * 1000 + M + N + K)
    dev = "cuda"
    # operands live inside wider buffers (leading dimension > width, odd offsets): strided views like q/k/v of qkv
    def view(rows, cols):
        buf = torch.randn(rows, cols + 5, device=dev)
        return buf[:, 3:3 + cols]
    A = view(M, K) if op in (0, 1) else view(K, M)
    B = view(N, K) if op in (0, 3) else view(K, N)
    out = torch.randn(M, N + 2, device=dev)[:, 1:1 + N]
    prev = out.clone()
    bias = torch.randn(N, device=dev)
    X.gemm(op, A, B, out, bias=bias, alpha=0.5, beta=2.0)
    a = A.double() if op in (0, 1) else A.double().t()
    b = B.double().t() if op in (0, 3) else B.double()
    ref = 0.5 * (a @ b) + bias.double() + 2.0 * prev.double()
    assert _rel(out, ref) < 2e-6, (op, M, N, K, _rel(out, ref))
    # aligned operands take the 16-byte load path: same answer
    A2, B2 = A.contiguous(), B.contiguous()
    out2 = torch.empty(M, N, device=dev)
    X.gemm(op, A2, B2, out2)
    assert _rel(out2, a @ b) < 2e-6


def test_xgemm_batched_attention_shapes():
    """QK^T, PV, and the three attention gradients as (batch, head)-strided problems on a fused qkv buffer."""
    from icka_amd import exact as X
    torch.manual_seed(7)
    B, h, S, R, dh = 3, 4, 37, 50, 16
    H = h * dh
    q = torch.randn(B * S, 3 * H, device="cuda")[:, :H]
    kv = torch.randn(B * R, 2 * H, device="cuda")
    k, v = kv[:, :H], kv[:, H:]
    P = torch.empty(B, h, S, R, device="cuda")
    X.gemm_raw(X.GEMM_NT, S, R, dh, q, q.stride(0), (S * q.stride(0), dh), k, k.stride(0), (R * k.stride(0), dh), P, R,
               (h * S * R, S * R), B, h)
    q4 = q.reshape(B, S, h, dh).permute(0, 2, 1, 3).double()
    k4 = k.reshape(B, R, h, dh).permute(0, 2, 1, 3).double()
    v4 = v.reshape(B, R, h, dh).permute(0, 2, 1, 3).double()
    assert _rel(P, q4 @ k4.transpose(-1, -2)) < 2e-6
    ctx = torch.empty(B * S, H, device="cuda")
    X.gemm_raw(X.GEMM_NN, S, dh, R, P, R, (h * S * R, S * R), v, v.stride(0), (R * v.stride(0), dh), ctx, H, (S * H, dh),
               B, h)
    ref = (P.double() @ v4).permute(0, 2, 1, 3).reshape(B * S, H)
    assert _rel(ctx, ref) < 2e-6
    dkv = torch.zeros(B * R, 2 * H, device="cuda")
    dv = dkv[:, H:]
    X.gemm_raw(X.GEMM_TN, R, dh, S, P, R, (h * S * R, S * R), ctx, H, (S * H, dh), dv, dv.stride(0), (R * dv.stride(0), dh),
               B, h)
    c4 = ctx.reshape(B, S, h, dh).permute(0, 2, 1, 3).double()
    ref = (P.double().transpose(-1, -2) @ c4).permute(0, 2, 1, 3).reshape(B * R, H)
    assert _rel(dv, ref) < 2e-6
    assert dkv[:, :H].abs().max().item() == 0.0      # nothing written outside the strided view


def _build(cfg, regions=49, variant="cl", max_seq_length=128):
    import icka_amd
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    c = BertConfig(cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                   num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                   max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"])
    m = MTCCMBertForMMTokenClassificationCRF(c, layer_num1=cfg["layer_num1"], num_labels=cfg["num_labels"],
                                             regions=regions, variant=variant, max_seq_length=max_seq_length)
    synth.fill_module_(m)
    return icka_amd.set_precision(m.cuda(), "fp32")


def _run(model, batch, labels=True):
    g = {k: v.cuda() for k, v in batch.items()}
    return model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"],
                 g["visual_embeds_mean"], g["visual_embeds_att"], labels=g["labels"] if labels else None)


@pytest.mark.parametrize("name", ["tiny_cl_r49", "tiny_cl_masks", "tiny_gatecl_s128", "base_cl_s64_r36",
                                  "base_cl_s128_r49"])
def test_fp32_mode_meets_1e3_on_reference_fixture(name):
    case = load_case(name)
    exp = case["expected"]
    model = _build(case["cfg"], case["cfg"]["regions"], variant=case["variant"],
                   max_seq_length=case["batch"]["input_ids"].shape[1]).eval()
    logits = _run(model, case["batch"], labels=False)
    assert logits.dtype == torch.float32 and tuple(logits.shape) == exp["logits"].shape
    err = np.abs(logits.detach().cpu().numpy() - exp["logits"]).max()
    model.zero_grad()
    loss = _run(model, case["batch"], labels=True)
    loss.backward()
    params = dict(model.named_parameters())
    gmax = float(exp["grad_norms"].max())
    worst, worst_key = 0.0, ""
    for n, gn in zip([str(x) for x in exp["grad_names"]], exp["grad_norms"]):
        if n not in params or gn == 0.0:
            continue
        g = params[n].grad
        assert g is not None, n
        rel = abs(g.norm().item() - gn) / (gn + 1e-6 * gmax)
        key = "grad/" + n
        if key in exp:
            ref = torch.from_numpy(exp[key])
            rel = max(rel, ((g.cpu() - ref).norm() / (ref.norm() + 1e-6 * gmax)).item())
        if rel > worst:
            worst, worst_key = rel, n
    print("\n[fp32 %s] logits max abs err %.3e (tol %.0e)  loss %.6f (ref %.6f)  worst grad rel %.3e at %s"
          % (name, err, FP32_TOL, loss.item(), float(exp["loss"][0]), worst, worst_key))
    assert err < FP32_TOL, "logits max abs err %.3e" % err
    assert abs(loss.item() - float(exp["loss"][0])) < FP32_TOL
    assert worst < GRAD_TOL, (worst_key, worst)
    assert model.bert.embeddings.word_embeddings.weight.grad[0].abs().max().item() == 0.0   # padding_idx row


def test_fp32_blocks_and_scalar_gate_against_fixture():
    """BertModel (all layers + pooler), a 2-layer BertCrossEncoder and the Cross_Modal scalar gate in fp32 mode."""
    import icka_amd
    from icka_amd.config import BertConfig
    from icka_amd.modeling import BertCrossEncoder, BertModel, cls_layer_both, scalar_gate_fusion
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
    m = icka_amd.set_precision(m.cuda().eval(), "fp32")
    ids, seg, msk = (torch.from_numpy(z[k]).cuda() for k in ("input_ids", "segment_ids", "input_mask"))
    layers, pooled = m.bert(ids, seg, msk)
    errs = [np.abs(l.detach().cpu().numpy() - z["layers"][i]).max() for i, l in enumerate(layers)]
    errs.append(np.abs(pooled.detach().cpu().numpy() - z["pooled"]).max())
    img = (1.0 - torch.from_numpy(z["added_attention_mask"])[:, :49].float())[:, None, None, :] * -10000.0
    cross = m.txt2img_attention(torch.from_numpy(z["layers"][-1]).cuda(), torch.from_numpy(z["s2"]).cuda(), img.cuda())
    errs += [np.abs(c.detach().cpu().numpy() - z["cross"][i]).max() for i, c in enumerate(cross)]

    class Gate(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.cls_layer = cls_layer_both(128, 128)
            self.aux_head = torch.nn.Linear(128, 1)

    g = Gate()
    synth.fill_module_(g)
    g = icka_amd.set_precision(g.cuda(), "fp32")
    cr = torch.from_numpy(z["cross"][-1]).cuda().requires_grad_(True)
    tok = torch.from_numpy(z["tok"]).cuda().requires_grad_(True)
    out = scalar_gate_fusion(g, cr, tok)
    errs.append(np.abs(out.detach().cpu().numpy() - z["blended"]).max())
    print("\n[fp32 blocks] max abs errs", ["%.2e" % e for e in errs])
    assert max(errs) < FP32_TOL
    out.sum().backward()
    c64 = cr.detach().double().requires_grad_(True)
    t64 = tok.detach().double().requires_grad_(True)
    P = {k: v.detach().double() for k, v in g.state_dict().items()}
    feat = torch.nn.functional.layer_norm(c64[:, 0] + t64[:, 0], (128,), P["cls_layer.proj_norm.weight"],
                                          P["cls_layer.proj_norm.bias"], 1e-5)
    gg = torch.sigmoid((feat @ P["cls_layer.proj.weight"].t() + P["cls_layer.proj.bias"]) @ P["aux_head.weight"].t()
                       + P["aux_head.bias"]).view(-1, 1, 1)
    (gg * t64 + (1 - gg) * c64).sum().backward()
    assert _rel(cr.grad, c64.grad) < 1e-4 and _rel(tok.grad, t64.grad) < 1e-4


def test_fp32_train_mode_is_seeded_and_consistent():
    """Dropout in fp32 mode: same seed -> identical loss and gradients; masks differ from eval."""
    case = load_case("tiny_cl_r49")
    model = _build(case["cfg"]).train()
    _run(model, case["batch"])           # builds the arena
    outs = []
    for _ in range(2):
        model._icka_arena.set_seed(99)
        model.zero_grad()
        loss = _run(model, case["batch"])
        loss.backward()
        outs.append((loss.item(), model.classifier.weight.grad.clone(), model.bert.embeddings.LayerNorm.weight.grad.clone()))
    assert outs[0][0] == outs[1][0]
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    model.eval()
    assert abs(_run(model, case["batch"]).item() - outs[0][0]) > 1e-6


def test_fp32_mode_other_head_size():
    """The fp32 mode has no head-size restriction (the bf16 kernels are built for 64): 96/4 = 24-wide heads vs oracle."""
    import icka_amd
    from icka_amd.config import BertConfig
    from icka_amd.modeling import BertModel
    from oracle import mner_oracle as O
    cfg = BertConfig(300, hidden_size=96, num_hidden_layers=2, num_attention_heads=4, intermediate_size=200,
                     max_position_embeddings=40)
    m = BertModel(cfg)
    synth.fill_module_(m)
    P = {"bert." + k: v.detach().clone() for k, v in m.state_dict().items()}
    m = icka_amd.set_precision(m.cuda().eval(), "fp32")
    b = synth.synthetic_batch(3, 33, 36, vocab_size=300, seed=5)
    layers, pooled = m(b["input_ids"].cuda(), b["segment_ids"].cuda(), b["input_mask"].cuda())
    ocfg = O.OracleConfig(vocab_size=300, hidden_size=96, num_hidden_layers=2, num_attention_heads=4, intermediate_size=200,
                          max_position_embeddings=40)
    ref_layers, ref_pooled = O.bert_model(P, "bert", b["input_ids"], b["segment_ids"], b["input_mask"], ocfg)
    assert (layers[-1].cpu() - ref_layers[-1]).abs().max().item() < FP32_TOL
    assert (pooled.cpu() - ref_pooled).abs().max().item() < FP32_TOL
    # since round 4 the 16-bit modes take this geometry too (attention core on the f32-input kernels, ops._attn_generic_fwd)
    l16, _ = icka_amd.set_precision(m, "bf16")(b["input_ids"].cuda(), b["segment_ids"].cuda(), b["input_mask"].cuda())
    assert (l16[-1].float().cpu() - ref_layers[-1]).abs().max().item() < 3e-2


# ----------------------------------------------------------------------------------------- SURVEY.md section 8(f) rows
@pytest.mark.parametrize("B,S,H", [(2, 5, 32), (4, 16, 64), (3, 40, 256), (8, 24, 1024)])
def test_fp32_bilstm_against_aten(B, S, H):
    """nn.LSTM on the CPU (the arithmetic the reference itself calls, :905-908) vs the fp32-mode BiLSTM."""
    import icka_amd
    from icka_amd.lstm import BiLSTM
    torch.manual_seed(B * 100 + S)
    ref = torch.nn.LSTM(H, H, batch_first=True, bidirectional=True)
    mine = BiLSTM(H, H)
    mine.load_state_dict(ref.state_dict())
    mine = icka_amd.set_precision(mine.cuda(), "fp32")
    x = torch.randn(B, S, H) * 0.5
    xg = x.cuda().requires_grad_(True)
    out, (h_n, c_n) = mine(xg)
    xr = x.clone().requires_grad_(True)
    ro, (rh, rc) = ref(xr)
    assert out.dtype == torch.float32
    err = (out.cpu() - ro).abs().max().item()
    assert err < 1e-5, err
    assert (h_n.cpu() - rh).abs().max().item() < 1e-5 and (c_n.cpu() - rc).abs().max().item() < 1e-5
    w = torch.randn(B, S, 2 * H, generator=torch.Generator().manual_seed(1))
    (out * w.cuda()).sum().backward()
    (ro * w).sum().backward()
    worst = _rel(xg.grad.cpu(), xr.grad)
    for n, p in mine.named_parameters():
        worst = max(worst, _rel(p.grad.cpu(), dict(ref.named_parameters())[n].grad))
    print("\n[fp32 BiLSTM B%d S%d H%d] max abs out err %.3e, worst gradient rel-L2 %.3e" % (B, S, H, err, worst))
    assert worst < 1e-4


def test_fp32_published_model_against_reference_fixture():
    """The reference's published model (Cross_Modal_Interaction_Module.py:887-1057) in fp32 mode: emissions of the
    fixture made by the reference's own forward within 1e-3, loss and every parameter gradient against the CPU oracle."""
    import os
    import icka_amd
    from test_cross_modal_cpu import build_case, oracle_emissions
    from test_cross_modal_gpu import _args
    from oracle import crf_oracle as OC
    fx = np.load(os.path.join(GOLDEN_DIR, "cross_modal_h1024_l1.npz"))
    model, ocfg, ocfg_r, b = build_case(fx)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = icka_amd.set_precision(model.cuda().eval(), "fp32")
    g = {k: v.cuda() for k, v in b.items()}
    em = model(**_args(g))
    ref = torch.from_numpy(fx["emissions"])
    err = (em.cpu() - ref).abs().max().item()
    loss = model(mode="train", **_args(g))
    loss.backward()
    oem, _ = oracle_emissions(P, ocfg, ocfg_r, b)
    mask = b["output_mask"].bool()
    crfP = [P["crf.start_transitions"], P["crf.end_transitions"], P["crf.transitions"]]
    rloss = -OC.crf_reduce(OC.crf_llh(oem, b["labels"], mask, *crfP), mask, "token_mean")
    rloss.backward()
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst, wkey, n = 0.0, "", 0
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        assert p.grad is not None, k
        rel = ((p.grad.cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-4 * gmax)).item()
        n += 1
        if rel > worst:
            worst, wkey = rel, k
    print("\n[fp32 published model] emissions vs reference fixture max abs err %.3e (tol %.0e), loss %.6f (oracle %.6f), worst "
          "gradient rel-L2 %.3e at %s over %d tensors" % (err, FP32_TOL, loss.item(), rloss.item(), worst, wkey, n))
    assert err < FP32_TOL
    assert abs(loss.item() - rloss.item()) < FP32_TOL
    assert n > 100 and worst < GRAD_TOL, (wkey, worst)
    assert model(mode="test", **_args(g)) == OC.crf_decode(em.cpu(), mask, *[p.detach() for p in crfP])


def test_fp32_gate1_tagger_against_cpu_composition():
    """`_gate_1` (trunk -> BiLSTM -> classifier -> CRF, :2383-2483) in fp32 mode vs trunk oracle + ATen nn.LSTM + CRF oracle."""
    import torch.nn.functional as F
    import icka_amd
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF_gate_1
    from oracle import crf_oracle as OC
    from oracle import mner_oracle as O
    B, S, H, C = 4, 32, 128, 13
    cfg = BertConfig(512, hidden_size=H, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF_gate_1(cfg, num_labels=C)
    synth.fill_module_(model)
    with torch.no_grad():
        for n, p in model.lstm.named_parameters():
            p.mul_(4.0)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = icka_amd.set_precision(model.cuda().eval(), "fp32")
    b = synth.synthetic_batch(B, S, 49, vocab_size=512, seed=3, layout="BCHW")
    g = {k: v.cuda() for k, v in b.items()}
    args = dict(input_ids=g["input_ids"], segment_ids=g["segment_ids"], input_mask=g["input_mask"],
                ori_input_ids=g["input_ids"], ori_input_mask=g["input_mask"], ori_segment_ids=g["segment_ids"],
                added_attention_mask=g["added_attention_mask"], visual_embeds_att=g["visual_embeds_att"],
                output_mask=g["input_mask"], labels=g["labels"])
    em = model(**args)
    loss = model(mode="train", **args)
    loss.backward()
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=H, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=64)
    _, cross, _ = O.mner_trunk(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                               b["visual_embeds_att"], 1, 49, False)
    lstm = torch.nn.LSTM(H, H, batch_first=True, bidirectional=True)
    x, _ = torch.func.functional_call(lstm, {k[len("lstm."):]: v for k, v in P.items() if k.startswith("lstm.")}, (cross,))
    ref_em = F.linear(x, P["classifier.weight"], P["classifier.bias"])
    mask = b["input_mask"].bool()
    crfP = [P["crf.start_transitions"], P["crf.end_transitions"], P["crf.transitions"]]
    rloss = -OC.crf_reduce(OC.crf_llh(ref_em, b["labels"], mask, *crfP), mask, "token_mean")
    rloss.backward()
    err = (em.cpu() - ref_em.detach()).abs().max().item()
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst = max(((p.grad.cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-4 * gmax)).item()
                for k, p in model.named_parameters() if P[k].grad is not None)
    print("\n[fp32 _gate_1 tagger] emissions max abs err %.3e, loss %.6f (oracle %.6f), worst gradient rel-L2 %.3e"
          % (err, loss.item(), rloss.item(), worst))
    assert err < FP32_TOL and abs(loss.item() - rloss.item()) < FP32_TOL and worst < GRAD_TOL
