"""GPU: icka_amd.optim.ArenaAdamW -- clip at the global gradient norm + AdamW over the flat arena buffers in three launches --
against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on clones of the same parameters and gradients (the reference's
update: My_cross_attention.py:743-751 groups, :831-844 clip / step / zero_grad; the transformers.AdamW class it imported is
absent from the installed transformers 5.x, which points to torch.optim.AdamW: parity with the 4.x class is unpinned)."""
import pytest
import torch

from icka_amd import synth

pytestmark = pytest.mark.gpu


def _model():
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    m = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(m)
    g = {k: v.cuda() for k, v in synth.synthetic_batch(4, 32, 36, vocab_size=512, seed=5).items()}
    args = (g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
            g["visual_embeds_att"])
    return m.cuda().eval(), args, g["labels"]


def test_arena_adamw_matches_torch_adamw_with_clipping(monkeypatch):
    from icka_amd import kernels as K
    from icka_amd.optim import ArenaAdamW, reference_param_groups
    model, args, labels = _model()
    model(*args, labels=labels).backward()            # builds the arena
    A = model._icka_arena
    A.shadow_policy = "tracked"
    opt = ArenaAdamW(model, lr=1e-2, weight_decay=0.01, max_grad_norm=0.05)     # (a norm small enough to clip for real)
    # reference copy: independent CPU-side clones driven by the SAME gradients
    names = [n for n, _ in model.named_parameters()]
    ref = {n: p.detach().clone().requires_grad_(True) for n, p in model.named_parameters()}
    groups = reference_param_groups(model, 0.01)
    idx = {id(p): n for n, p in model.named_parameters()}
    topt = torch.optim.AdamW([{"params": [ref[idx[id(p)]] for p in g["params"]], "weight_decay": g["weight_decay"]}
                              for g in groups], lr=1e-2)
    assert {nd for nd in ("bias", "LayerNorm") if any(nd in idx[id(p)] for p in groups[1]["params"])} == {"bias", "LayerNorm"}
    casts = []
    real = K.cast_f32_to_bf16
    monkeypatch.setattr(K, "cast_f32_to_bf16", lambda s, d: (casts.append(s.numel()), real(s, d))[1])
    worst = 0.0
    for it in range(4):
        model.zero_grad()
        for r in ref.values():
            r.grad = None
        model(*args, labels=labels).backward()
        with_grad = [n for n in names if dict(model.named_parameters())[n].grad is not None]
        for n, p in model.named_parameters():
            if p.grad is not None:
                ref[n].grad = p.grad.detach().clone()
        tn = torch.nn.utils.clip_grad_norm_([ref[n] for n in with_grad], 0.05)
        topt.step()
        opt.step()
        torch.cuda.synchronize()
        assert abs(opt.grad_norm().item() - tn.item()) < 1e-5 * tn.item()
        for n, p in model.named_parameters():
            d = (p.detach() - ref[n].detach()).abs().max().item()
            worst = max(worst, d / (ref[n].detach().abs().max().item() + 1e-12))
            assert d <= 2e-6 * max(1.0, ref[n].detach().abs().max().item()), (it, n, d)
        # the update kernel wrote the bf16 shadow of everything it changed
        for lo, hi in A._cast_ranges:
            assert torch.equal(A.shadow[lo:hi], A.flat[lo:hi].to(torch.bfloat16)), it
    # "tracked" policy: after the first forward no arena re-cast ran -- the optimizer leaves fresh shadows behind
    assert len(casts) == 0, casts
    print("\n[ArenaAdamW vs clip_grad_norm_ + torch.optim.AdamW, 4 steps] worst relative parameter difference %.2e" % worst)


def test_arena_adamw_follows_a_lambda_lr_schedule_and_skips_parameters_without_gradient():
    from icka_amd.optim import ArenaAdamW
    model, args, labels = _model()
    model(*args, labels=labels).backward()
    opt = ArenaAdamW(model, lr=1e-3, max_grad_norm=1.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: 0.5 ** s)       # get_linear_schedule_with_warmup is a LambdaLR
    pooler_before = model.bert.pooler.dense.weight.detach().clone()          # the heads never use the pooler: no gradient
    w_before = model.classifier.weight.detach().clone()
    opt.step(); sched.step()
    assert abs(opt.param_groups[0]["lr"] - 0.5e-3) < 1e-12
    torch.cuda.synchronize()
    assert torch.equal(model.bert.pooler.dense.weight.detach(), pooler_before)
    assert not torch.equal(model.classifier.weight.detach(), w_before)


def test_arena_adamw_state_dict_resumes_moments_and_step_count():
    """the moments and the step count live outside Optimizer.state; state_dict() / load_state_dict() carry them, so a
    resumed run (new model object + new optimizer, state loaded BEFORE the first step binds the arena) continues exactly like
    the uninterrupted one; a foreign layout and a plain torch state_dict are refused; a rebuilt arena after step 1 raises."""
    import copy
    from icka_amd.optim import ArenaAdamW
    model, args, labels = _model()
    twin = copy.deepcopy(model)

    def run(m, opt, n):
        for _ in range(n):
            m.zero_grad()
            m(*args, labels=labels).backward()
            opt.step()
        torch.cuda.synchronize()

    full = ArenaAdamW(model, lr=1e-2, max_grad_norm=1.0)
    run(model, full, 4)
    first = ArenaAdamW(twin, lr=1e-2, max_grad_norm=1.0)
    run(twin, first, 2)
    sd = first.state_dict()
    assert sd["icka_t"] == 2 and sd["icka_m"].abs().sum().item() > 0
    resumed_model = copy.deepcopy(twin)
    resumed_model._icka_arena = None                 # a fresh process would build its own arena on the first forward
    for mod in resumed_model.modules():
        object.__setattr__(mod, "_icka_arena", None)
    second = ArenaAdamW(resumed_model, lr=1e-2, max_grad_norm=1.0)
    second.load_state_dict(sd)                       # before any step: installed when the arena is bound
    run(resumed_model, second, 2)
    worst = max((p.detach() - q.detach()).abs().max().item() for p, q in zip(model.parameters(), resumed_model.parameters()))
    assert worst < 1e-6, worst
    assert second.state_dict()["icka_t"] == 4
    with pytest.raises(ValueError, match="icka_t"):
        ArenaAdamW(model, lr=1e-2).load_state_dict(torch.optim.AdamW(model.parameters()).state_dict())
    bad = dict(sd)
    bad["icka_layout"] = [("x", 0, 8)]
    with pytest.raises(ValueError, match="another parameter layout"):
        full.load_state_dict(bad)
    # arena rebuilt behind a stepping optimizer
    model.float()                                    # no-op cast keeps the views; force a rebuild the way .to() does
    for mod in model.modules():
        object.__setattr__(mod, "_icka_arena", None)
    with torch.no_grad():
        for p in model.parameters():
            p.data = p.data.clone()
    model.zero_grad()
    model(*args, labels=labels).backward()
    with pytest.raises(RuntimeError, match="rebuilt"):
        full.step()
