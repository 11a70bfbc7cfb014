"""GPU: the reference's WHOLE loop through the capture wrappers, not only its steady state.

My_cross_attention.py: the train loader has no ``drop_last`` (:708) -- the last batch of an epoch is short; after every epoch the
dev pass runs in ``model.eval()`` under ``torch.no_grad()`` at ``eval_batch_size`` (:734, :846-875); then ``model.train()`` again.
Gradient accumulation (:821-822, :831) makes the short batch arrive in the MIDDLE of an accumulation cycle.  Two epochs of that
loop on a tiny configuration, launched eagerly and through ``graph.GraphedModule`` (loop body unchanged) and ``graph.GraphedStep``:
same losses, same dev logits, same parameters, no exception, at most 3 captures.  Dropout probabilities are 0 so that train mode
is deterministic (the kernels of train mode still run); a p = 0.1 pass checks that nothing raises and the loss stays finite.
Bars: the two runs launch the same kernels on the same data; what differs is the order of the f32 atomics of the embedding scatter
and the split-K classifier gradient, which AdamW's m / sqrt(v) amplifies over the four updates (measured 9e-6 in fp32 mode)."""
import copy

import pytest
import torch

import icka_amd
from icka_amd import synth

pytestmark = pytest.mark.gpu

NAMES = ("input_ids", "segment_ids", "input_mask", "added_attention_mask", "visual_embeds_mean", "visual_embeds_att", "labels")
K_ACC = 2


def _model(precision, p_drop=0.0):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64, hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    m = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(m)
    return icka_amd.set_precision(m.cuda().train(), precision)


def _batch(n, seed):
    b = synth.synthetic_batch(n, 32, 36, vocab_size=512, seed=seed)
    return tuple(b[k].cuda() for k in NAMES)


def _data():
    train = [_batch(4, 200), _batch(4, 201), _batch(4, 202), _batch(2, 203)]     # 14 examples at train_batch_size 4, no drop_last
    dev = [_batch(3, 300), _batch(3, 301)]                                       # eval_batch_size 3
    return train, dev


def _loop(model, forward, train, dev, epochs=2):
    """My_cross_attention.py:790-875 around ``forward(*inputs, labels=) -> loss`` / ``forward(*inputs) -> logits``."""
    from icka_amd.optim import reference_param_groups
    opt = torch.optim.AdamW(reference_param_groups(model, 0.01), lr=1e-3)
    losses, logits = [], []
    for _ in range(epochs):
        model.train()
        model.zero_grad()
        for step, b in enumerate(train):
            loss = forward(*b[:6], labels=b[6]) / K_ACC          # :814-822
            loss.backward()                                      # :827
            losses.append(loss.item())
            if (step + 1) % K_ACC == 0:                          # :831-844
                torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
                opt.step()
                model.zero_grad()
        model.eval()                                             # :846
        for b in dev:
            with torch.no_grad():                                # :870-875
                logits.append(forward(*b[:6]).float().cpu().clone())
    torch.cuda.synchronize()
    return losses, logits, {n: p.detach().clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("precision,bar", [("fp32", 3e-5), ("bf16", 1e-3)])
def test_graphed_module_runs_the_reference_loop_with_short_last_batch_and_dev_pass(precision, bar):
    from icka_amd.graph import GraphedModule
    base = _model(precision)
    train, dev = _data()
    eager_model = copy.deepcopy(base)
    le, ge, pe = _loop(eager_model, eager_model, train, dev)
    model = copy.deepcopy(base)
    gm = GraphedModule(model, train[0][:6], {"labels": train[0][6]})
    lg, gg, pg = _loop(gm, gm, train, dev)                       # the wrapper stands in for the module everywhere (.train(), ...)
    print("\n[%s, GraphedModule, whole loop] losses %s; %s" % (precision, ["%.5f" % x for x in lg], gm.stats))
    assert gm.captures <= 3 and gm.stats["eager_calls"] == 0, gm.stats
    for a, b in zip(le, lg):
        assert abs(a - b) <= bar * max(1.0, abs(a)), (le, lg)
    for a, b in zip(ge, gg):
        assert (a - b).abs().max().item() <= 20 * bar, (a - b).abs().max().item()
    worst = max((pe[n] - pg[n]).abs().max().item() / (pe[n].abs().max().item() + 1e-6) for n in pe)
    print("    worst parameter difference vs the eager loop after two epochs: %.3e (bar %.1e)" % (worst, bar))
    assert worst <= bar, worst
    # with a cache of ONE entry everything but the full training batch runs eagerly: same numbers, still no exception
    model1 = copy.deepcopy(base)
    gm1 = GraphedModule(model1, train[0][:6], {"labels": train[0][6]}, max_captures=1)
    l1, g1, p1 = _loop(gm1, gm1, train, dev)
    assert gm1.captures == 1 and gm1.stats["eager_calls"] == 2 * (1 + len(dev)), gm1.stats
    for a, b in zip(le, l1):
        assert abs(a - b) <= bar * max(1.0, abs(a)), (le, l1)
    assert max((pe[n] - p1[n]).abs().max().item() / (pe[n].abs().max().item() + 1e-6) for n in pe) <= bar
    gm.close()
    gm1.close()


@pytest.mark.parametrize("precision,bar", [("fp32", 3e-5), ("bf16", 1e-3)])
def test_graphed_step_runs_the_reference_loop_with_short_last_batch_and_dev_pass(precision, bar):
    """The one-graph form: ``gs(*batch)`` = forward + backward.  The short batch gets a capture of its own in the middle of an
    accumulation cycle (the gradients held are put aside during its warm-up and restored); the dev pass calls the module itself
    (a step function has no forward-only form)."""
    from icka_amd.graph import GraphedStep
    base = _model(precision)
    train, dev = _data()
    eager_model = copy.deepcopy(base)
    le, ge, pe = _loop(eager_model, eager_model, train, dev)
    model = copy.deepcopy(base)

    def micro(ids, seg, mask, added, vmean, vatt, labels):
        loss = model(ids, seg, mask, added, vmean, vatt, labels=labels) / K_ACC
        loss.backward()
        return loss

    gs = GraphedStep(model, micro, inputs=train[0])

    class _Loss(object):          # adapts gs to the loop's  loss = forward(...) / K_ACC; loss.backward()  shape
        def __init__(self, t):
            self.t = t

        def __truediv__(self, k):
            return self

        def backward(self):
            pass

        def item(self):
            return self.t.item()

    def forward(*args, labels=None):
        if labels is None:
            return model(*args)
        return _Loss(gs(*args, labels))

    lg, gg, pg = _loop(model, forward, train, dev)
    print("\n[%s, GraphedStep, whole loop] losses %s; %s" % (precision, ["%.5f" % x for x in lg], gs.stats))
    assert gs.captures == 2 and gs.stats["eager_calls"] == 0, gs.stats
    for a, b in zip(le, lg):
        assert abs(a - b) <= bar * max(1.0, abs(a)), (le, lg)
    worst = max((pe[n] - pg[n]).abs().max().item() / (pe[n].abs().max().item() + 1e-6) for n in pe)
    assert worst <= bar, worst
    gs.close()


def test_whole_loop_with_dropout_active_stays_finite_and_never_raises():
    from icka_amd.graph import GraphedModule
    model = _model("bf16", p_drop=0.1)
    train, dev = _data()
    gm = GraphedModule(model, train[0][:6], {"labels": train[0][6]})
    losses, logits, _ = _loop(gm, gm, train, dev)
    assert all(l == l and abs(l) < 1e3 for l in losses) and all(torch.isfinite(x).all() for x in logits)
    assert gm.captures <= 3 and losses[0] != losses[4]          # the second epoch's weights moved
    gm.close()


def test_north_star_keyword_signature():
    """BASELINE.json north_star: ``forward(input_ids, attention_mask, visual_feats, ...)`` -- keyword aliases of the reference's
    positional form (gate_cl_modeling.py:1319-1320); omitted arguments take the values the reference's feature builder gives
    them (segment ids 0, added_attention_mask = ones[R] + text mask, My_cross_attention.py:362, :373)."""
    model = _model("bf16").eval()
    b = _batch(2, 7)
    want = model(b[0], b[1], b[2], b[3], b[4], b[5])
    got = model(b[0], attention_mask=b[2], visual_feats=b[5])
    assert torch.equal(want, got)
    assert torch.equal(model(b[0], token_type_ids=b[1], attention_mask=b[2], visual_feats=b[5], labels=b[6]),
                       model(b[0], b[1], b[2], b[3], b[4], b[5], labels=b[6]))
    with pytest.raises(TypeError):
        model(b[0], b[1], b[2], b[3], b[4], b[5], visual_feats=b[5])
    with pytest.raises(TypeError):
        model(b[0], attention_mask=b[2])
