"""GPU: the reference's OWN optimisation recipe through the captured step.

My_cross_attention.py:797-844 per optimisation step: ``gradient_accumulation_steps`` (default 5, :587-590) micro-batches of
DIFFERENT inputs, each moved to the device (:797-798), loss / 5 (:821-822), backward (:827), then clip_grad_norm_(1.0)
(:841), optimizer.step() (AdamW over the two weight-decay groups of :743-751), scheduler.step() (linear warm-up, :756-757)
and model.zero_grad() (:842-844).  Here that loop runs twice on copies of one seeded model -- launched eagerly, and through
``GraphedStep(model, step_fn, inputs=first_batch)`` with ``gs(*batch)`` per micro-batch -- and the loss sequence and every
parameter after two optimisation steps are compared (dropout off: eval mode).  fp32 mode: the two runs launch the same
kernels on the same data, bar 1e-6 (the embedding scatter's f32 atomics are order-dependent in the last bit); bf16: same bar
class, one bf16 rounding of slack."""
import copy

import pytest
import torch

import icka_amd
from icka_amd import synth

pytestmark = pytest.mark.gpu

K_ACC = 5          # gradient_accumulation_steps, My_cross_attention.py:587-590
OPT_STEPS = 2
NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")


def _model(precision):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    m = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(m)
    return icka_amd.set_precision(m.cuda().eval(), precision)


def _batches(n, device="cuda"):
    out = []
    for i in range(n):
        b = synth.synthetic_batch(4, 32, 36, vocab_size=512, seed=100 + i)
        out.append(tuple(b[k].to(device) for k in NAMES))
    return out


def _recipe(model, run_micro, batches, use_arena_adamw):
    """The reference's loop (:797-844) around ``run_micro(batch) -> loss``."""
    from icka_amd.optim import ArenaAdamW, reference_param_groups
    total = OPT_STEPS
    if use_arena_adamw:
        opt = ArenaAdamW(model, lr=1e-3, weight_decay=0.01, max_grad_norm=1.0)
    else:
        opt = torch.optim.AdamW(reference_param_groups(model, 0.01), lr=1e-3)
    warm = 1      # get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps) is this LambdaLR
    sched = torch.optim.lr_scheduler.LambdaLR(
        opt, lambda s: float(s) / max(1, warm) if s < warm else max(0.0, float(total + 1 - s) / max(1, total + 1 - warm)))
    model.zero_grad()
    losses = []
    for step, batch in enumerate(batches):
        loss = run_micro(batch)
        losses.append(loss.item())
        if (step + 1) % K_ACC == 0:
            if not use_arena_adamw:
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            sched.step()
            model.zero_grad()
    torch.cuda.synchronize()
    return losses, {n: p.detach().clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("arena_adamw", [False, True], ids=["torch-adamw", "arena-adamw"])
@pytest.mark.parametrize("precision,bar", [("fp32", 2e-6), ("bf16", 2e-4)])
def test_reference_recipe_eager_equals_graphed(precision, bar, arena_adamw):
    from icka_amd.graph import GraphedStep
    base = _model(precision)
    batches = _batches(K_ACC * OPT_STEPS)
    results = []
    for graphed in (False, True):
        model = copy.deepcopy(base)

        def micro(ids, seg, mask, added, vmean, vatt, labels):
            loss = model(ids, seg, mask, added, vmean, vatt, labels=labels) / K_ACC      # :821-822
            loss.backward()
            return loss

        if graphed:
            gs = GraphedStep(model, micro, inputs=batches[0])
            run = lambda b: gs(*b)          # noqa: E731  (new tensors every call: copied into the static buffers)
        else:
            run = lambda b: micro(*b)       # noqa: E731
        results.append(_recipe(model, run, batches, arena_adamw))
        if graphed:
            assert set(gs._graphs) == {False, True}, "one overwrite capture + one accumulate capture"
            gs.close()
    (le, pe), (lg, pg) = results
    print("\n[%s, %s] losses eager %s\n%s graphed %s" % (precision, "ArenaAdamW" if arena_adamw else "torch AdamW",
                                                       ["%.6f" % x for x in le], " " * 12, ["%.6f" % x for x in lg]))
    assert le[K_ACC] != le[0] and abs(le[-1] - le[0]) > 1e-5      # different inputs, and the weights moved
    for a, b in zip(le, lg):
        assert abs(a - b) <= bar * max(1.0, abs(a)), (le, lg)
    worst, wkey = 0.0, ""
    for n in pe:
        d = (pe[n] - pg[n]).abs().max().item() / (pe[n].abs().max().item() + 1e-6)
        if d > worst:
            worst, wkey = d, n
    print("    worst parameter difference after %d optimisation steps: %.3e (%s), bar %.1e" % (OPT_STEPS, worst, wkey, bar))
    assert worst <= bar, (worst, wkey)


def test_accumulated_gradients_equal_the_sum_of_micro_batches():
    """5 replays between two zero_grads == the eager sum of the 5 micro-batch gradients (fp32 mode), with host-resident
    batches (the reference moves every batch to the device inside the loop) and zero_grad(set_to_none=False) in between."""
    from icka_amd.graph import GraphedStep
    model = _model("fp32")
    batches = _batches(K_ACC, device="cpu")
    dev_batches = [tuple(t.cuda() for t in b) for b in batches]

    def micro(ids, seg, mask, added, vmean, vatt, labels):
        loss = model(ids, seg, mask, added, vmean, vatt, labels=labels) / K_ACC
        loss.backward()
        return loss

    model.zero_grad()
    for b in dev_batches:
        micro(*b)
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    gs = GraphedStep(model, micro, inputs=dev_batches[0])
    for cycle in range(2):
        if cycle == 0:
            model.zero_grad()                   # set_to_none=True: the first replay of the cycle overwrites
        else:
            model.zero_grad(set_to_none=False)  # zeros kept: the first replay accumulates onto them
        for b in batches:
            gs(*b)                              # pageable host tensors: copy_ stages them
        torch.cuda.synchronize()
        for n, p in model.named_parameters():
            if n in ref:
                assert p.grad is not None, n
                err = (p.grad - ref[n]).abs().max().item() / (ref[n].abs().max().item() + 1e-12)
                assert err < 2e-6, (cycle, n, err)
    gs.close()


def test_other_shapes_are_captured_on_first_sight_and_arity_is_checked():
    """A batch of another shape (the short last batch of an epoch, My_cross_attention.py:708) gets a capture of its own the first
    time it is seen and replays it afterwards; past ``max_captures`` the call runs ``step_fn`` eagerly -- in both cases the
    loss and the gradients of the eager step, never an exception.  Wrong arity stays a TypeError (a wrong call, not a new shape)."""
    from icka_amd.graph import GraphedStep
    model = _model("bf16")
    b = _batches(1)[0]

    def micro(ids, seg, mask, added, vmean, vatt, labels):
        loss = model(ids, seg, mask, added, vmean, vatt, labels=labels)
        loss.backward()
        return loss

    small = tuple(t[:2].contiguous() for t in b)
    model.zero_grad()
    ref = micro(*small).item()
    gref = model.classifier.weight.grad.clone()
    gs = GraphedStep(model, micro, inputs=b, warmup=1, max_captures=2)
    model.zero_grad()
    assert gs(*small).item() == pytest.approx(ref, rel=1e-6)          # first sight: captured (2 of 2)
    assert torch.allclose(model.classifier.weight.grad, gref, rtol=1e-4, atol=1e-7)
    model.zero_grad()
    assert gs(*small).item() == pytest.approx(ref, rel=1e-6) and gs.stats["captures"] == 2 and gs.captures == 2
    one = tuple(t[:1].contiguous() for t in b)
    model.zero_grad()
    l_one = gs(*one).item()                                           # a third shape: past the cap -> eager
    assert gs.stats["eager_calls"] == 1 and gs.captures == 2
    model.zero_grad()
    assert micro(*one).item() == pytest.approx(l_one, rel=1e-6)
    with pytest.raises(TypeError):
        gs(*b[:-1])
    model.zero_grad()
    l0 = gs(*b).item()
    l1 = gs(b).item()                 # a single tuple is unpacked
    assert l0 == l1
    no_inputs = GraphedStep(model, lambda: micro(*b), warmup=1)
    with pytest.raises(TypeError, match="without inputs"):
        no_inputs(*b)
    # keyword form
    d = dict(zip(NAMES, b))

    def micro_kw(**kw):
        return micro(*(kw[k] for k in NAMES))

    gk = GraphedStep(model, micro_kw, inputs=d, warmup=1)
    model.zero_grad()
    assert gk(**d).item() == l0
    with pytest.raises(TypeError):
        gk(*b)
    for g in (gs, no_inputs, gk):
        g.close()


@pytest.mark.parametrize("precision,bar", [("fp32", 2e-6), ("bf16", 2e-4)])
def test_graphed_module_keeps_the_reference_loop_body(precision, bar):
    """graph.GraphedModule: the reference's two lines -- ``loss = model(...)`` (:814-817) and ``loss.backward()`` (:827) -- stay as
    they are and replay a forward and a backward hipGraph.  The recipe of this file (5 micro-batches of different inputs,
    loss / 5 applied OUTSIDE the wrapper, clip, AdamW, schedule, zero_grad) eager vs wrapped: same losses and parameters; a
    no_grad call replays the forward only; zero_grad between forward and backward is honoured; the other train / eval mode and
    another call signature get captures of their own."""
    from icka_amd.graph import GraphedModule
    base = _model(precision)
    batches = _batches(K_ACC * OPT_STEPS)
    results = []
    for wrapped in (False, True):
        model = copy.deepcopy(base)
        call = GraphedModule(model, batches[0][:6], {"labels": batches[0][6]}) if wrapped else model

        def run(b):
            loss = call(*b[:6], labels=b[6]) / K_ACC            # the reference's own lines: forward ...
            loss.backward()                                     # ... and backward
            return loss.detach().clone()

        results.append(_recipe(model, run, batches, False))
        if wrapped:
            gm = call
    (le, pe), (lg, pg) = results
    for a, b in zip(le, lg):
        assert abs(a - b) <= bar * max(1.0, abs(a)), (le, lg)
    worst = max((pe[n] - pg[n]).abs().max().item() / (pe[n].abs().max().item() + 1e-6) for n in pe)
    print("\n[%s, GraphedModule] losses %s; worst parameter difference vs the eager loop %.3e" % (precision, ["%.6f" % x for x in lg], worst))
    assert worst <= bar, worst
    b = batches[0]
    with torch.no_grad():
        l0 = gm(*b[:6], labels=b[6]).item()
    assert abs(l0 - gm.model(*b[:6], labels=b[6]).item()) <= bar * max(1.0, abs(l0))
    # loss = model(x); optimizer.zero_grad(); loss.backward()  -> the backward overwrites
    gm.model.zero_grad()
    gm(*b[:6], labels=b[6]).backward()
    g1 = gm.model.classifier.weight.grad.clone()
    loss = gm(*b[:6], labels=b[6])
    gm.model.zero_grad()
    loss.backward()
    assert torch.equal(gm.model.classifier.weight.grad, g1)
    loss = gm(*b[:6], labels=b[6])
    loss.backward()                                             # gradients still held: accumulates
    assert torch.allclose(gm.model.classifier.weight.grad, 2 * g1, rtol=1e-3, atol=1e-7)
    # another call signature (no labels -> logits) and the other train / eval mode: captures of their own, no exception
    n0 = gm.captures
    logits = gm(*b[:6])
    assert logits.shape == (4, 32, 13) and gm.captures == n0 + 1
    assert torch.allclose(logits, gm.model(*b[:6]), rtol=0, atol=bar * 10)
    gm.model.train()
    lt = gm(*b[:6], labels=b[6])
    assert torch.isfinite(lt).item() and gm.captures == n0 + 2
    gm.close()


@pytest.mark.parametrize("pinned", [False, True], ids=["pageable", "pinned"])
def test_device_prefetcher_feeds_the_captured_step_in_order(pinned):
    """graph.DevicePrefetcher: host batches (the reference's loader hands over host tensors, My_cross_attention.py:795-798) arrive
    as device tuples one step ahead, in order, bit for bit, through two rotating buffer sets; a captured step fed from it gives the
    losses of the same step fed the device batches directly; arity / shape changes raise."""
    from icka_amd.graph import DevicePrefetcher, GraphedStep
    host = _batches(7, device="cpu")
    if pinned:
        host = [tuple(t.pin_memory() for t in b) for b in host]
    seen = 0
    for i, b in enumerate(DevicePrefetcher(host, "cuda")):
        assert all(t.is_cuda for t in b)
        for got, want in zip(b, host[i]):
            assert torch.equal(got.cpu(), want), i
        seen += 1
    assert seen == len(host) and len(DevicePrefetcher(host, "cuda")) == len(host)
    assert list(DevicePrefetcher([], "cuda")) == []
    model = _model("bf16")

    def micro(ids, seg, mask, added, vmean, vatt, labels):
        loss = model(ids, seg, mask, added, vmean, vatt, labels=labels)
        loss.backward()
        return loss

    gs = GraphedStep(model, micro, inputs=tuple(t.cuda() for t in host[0]), warmup=1)
    direct = []
    for b in host:
        model.zero_grad()
        direct.append(gs(*(t.cuda() for t in b)).item())
    fed = []
    for b in DevicePrefetcher(host, "cuda", depth=3):
        model.zero_grad()
        fed.append(gs(*b).item())
    assert fed == direct and len(set(fed)) > 1
    gs.close()
    # the short last batch of an epoch (no drop_last, :708) passes through with buffers of its own
    ragged = [host[0], tuple(t[:2] for t in host[1]), host[2]]
    for got, want in zip(DevicePrefetcher(ragged, "cuda"), ragged):
        assert all(torch.equal(g.cpu(), w) for g, w in zip(got, want))
    with pytest.raises(ValueError, match="arity"):
        list(DevicePrefetcher([host[0], host[1][:-1]], "cuda"))
    with pytest.raises(ValueError):
        DevicePrefetcher(host, "cuda", depth=1)
