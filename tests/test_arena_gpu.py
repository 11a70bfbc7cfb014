"""GPU: parameter / gradient bookkeeping around the kernels: the bf16 shadow follows parameter updates
made through ``.data`` (the reference's BertAdam, my_bert/optimization.py:153), zero_grad between forward and backward,
hipGraph replay with the reference loop's zero_grad-per-step (My_cross_attention.py:843)."""
import pytest
import torch

from icka_amd import synth
from golden_util import load_case

pytestmark = pytest.mark.gpu


def _model():
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    case = load_case("tiny_cl_r49")
    cfg = case["cfg"]
    c = BertConfig(cfg["vocab_size"], hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                   num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                   max_position_embeddings=cfg["max_position_embeddings"], type_vocab_size=cfg["type_vocab_size"])
    m = MTCCMBertForMMTokenClassificationCRF(c, layer_num1=cfg["layer_num1"], num_labels=cfg["num_labels"])
    synth.fill_module_(m)
    g = {k: v.cuda() for k, v in case["batch"].items()}
    args = (g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
            g["visual_embeds_att"])
    return m.cuda().eval(), args, g["labels"]


def _shadow_ok(A):
    """every GEMM operand range of the bf16 shadow equals the f32 master (embedding tables have no shadow)"""
    torch.cuda.synchronize()
    assert sum(hi - lo for lo, hi in A._cast_ranges) + sum(s.numel for s in A.order if s.is_table) >= A.total - 8 * len(A.order)
    return all(torch.equal(A.shadow[lo:hi], A.flat[lo:hi].to(torch.bfloat16)) for lo, hi in A._cast_ranges)


def test_shadow_follows_data_updates_like_bertadam():
    """p.data.add_() leaves every version counter untouched; the default policy re-casts in each outermost forward."""
    model, args, labels = _model()
    model(*args, labels=labels).backward()
    A = model._icka_arena
    assert A.shadow_policy == "always" and _shadow_ok(A)
    l0 = model(*args, labels=labels).item()
    v = sum(p._version for p in model.parameters())
    with torch.no_grad():
        for p in model.parameters():           # my_bert/optimization.py:153  p.data.add_(-update_with_lr)
            if p.grad is not None:
                p.data.add_(-0.05 * p.grad.data)
    assert sum(p._version for p in model.parameters()) == v      # invisible to the counters (the ADVICE scenario)
    l1 = model(*args, labels=labels).item()
    assert _shadow_ok(A), "GEMMs would keep running on stale weights"
    assert l1 < l0 - 1e-3                       # a gradient step on the same batch lowers the loss
    # opt-in tracked policy: in-place ops on the parameter / optimizer steps / mark_dirty() are seen, .data writes are not
    A.shadow_policy = "tracked"
    model(*args, labels=labels)
    with torch.no_grad():
        model.classifier.weight.mul_(1.5)       # bumps _version
    model(*args, labels=labels)
    assert _shadow_ok(A)
    model.classifier.weight.data.mul_(0.5)
    A.mark_dirty()                              # the documented escape hatch for .data writes under "tracked"
    model(*args, labels=labels)
    assert _shadow_ok(A)
    opt = torch.optim.SGD(model.parameters(), lr=0.01)
    model(*args, labels=labels).backward()
    opt.step()
    model(*args, labels=labels)
    assert _shadow_ok(A)


def test_zero_grad_between_forward_and_backward():
    model, args, labels = _model()
    model(*args, labels=labels).backward()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    for _ in range(2):                          # loss = model(x); optimizer.zero_grad(); loss.backward()
        loss = model(*args, labels=labels)
        opt.zero_grad()
        loss.backward()
    for n, p in model.named_parameters():
        if n in ref:
            assert torch.equal(p.grad, ref[n]), "%s accumulated across steps" % n


def test_parameters_outside_the_optimizer_accumulate_like_torch():
    model, args, labels = _model()
    head = [model.classifier.weight, model.classifier.bias]
    opt = torch.optim.SGD(head, lr=0.0)         # trunk frozen out of the optimizer: nobody clears its .grad
    model(*args, labels=labels).backward()
    g_head, g_trunk = head[0].grad.clone(), model.vismap2text.weight.grad.clone()
    opt.zero_grad()
    model(*args, labels=labels).backward()
    assert torch.equal(head[0].grad, g_head)
    assert torch.allclose(model.vismap2text.weight.grad, 2 * g_trunk, rtol=1e-3, atol=1e-7)


def test_graphed_step_with_reference_training_loop():
    from icka_amd import kernels as K
    from icka_amd.graph import GraphedStep
    model, args, labels = _model()     # eval mode: the loss sequence below is deterministic

    def step():
        loss = model(*args, labels=labels)
        loss.backward()
        return loss

    gs = GraphedStep(model, step)
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    losses = []
    for _ in range(4):
        loss = gs()
        assert model.classifier.weight.grad is not None and model.vismap2text.weight.grad is not None
        w0 = model.classifier.weight.detach().clone()
        opt.step()
        assert not torch.equal(w0, model.classifier.weight.detach()), "optimizer saw no gradient after replay"
        model.zero_grad()                       # My_cross_attention.py:843 (set_to_none=True is the default)
        assert model.classifier.weight.grad is None
        losses.append(loss.item())
    assert losses[-1] < losses[0]               # the replayed forward sees the updated weights
    assert K._NONCE_PTR == gs.nonce.data_ptr()
    gs.close()
    assert K._NONCE_PTR is None                 # kernels no longer point at memory the GraphedStep owned
    with pytest.raises(RuntimeError):
        gs()


def test_mixed16_graphed_training_loop_matches_eager_and_refreshes_both_shadows():
    """mixed16 under a captured step: the replayed forward reads the fp16 AND bf16 weight shadows of the weights the
    optimizer just wrote (through ``.data``-style in-place updates as well), and the loss sequence equals the eager one."""
    import copy
    import icka_amd
    from icka_amd.graph import GraphedStep
    base, args, labels = _model()
    seqs = []
    for graphed in (False, True):
        model = icka_amd.set_precision(copy.deepcopy(base), "mixed16")

        def step():
            loss = model(*args, labels=labels)
            loss.backward()
            return loss

        run = GraphedStep(model, step) if graphed else step
        opt = torch.optim.SGD(model.parameters(), lr=0.05)
        losses = []
        for i in range(4):
            if not graphed:
                model.zero_grad()
            loss = run()
            losses.append(loss.item())
            opt.step()
            if i == 1:      # an update that leaves no trace in any version counter
                with torch.no_grad():
                    model.classifier.weight.data.mul_(0.5)
            model.zero_grad()
        A = model._icka_arena
        torch.cuda.synchronize()
        A.sync()
        w = model.bert.encoder.layer[0].output.dense.weight
        assert torch.equal(A.w16(w), w.detach().to(torch.float16)) and torch.equal(A.w(w), w.detach().to(torch.bfloat16))
        seqs.append(losses)
        if graphed:
            run.close()
    assert seqs[0][-1] < seqs[0][0]
    for a, b in zip(*seqs):
        assert abs(a - b) < 2e-3 * max(1.0, abs(a)), seqs


def test_standalone_container_block_sees_weight_updates_between_calls():
    """BertAttention / BertCrossAttention launch nothing themselves; when one of them is the OUTERMOST call its
    children run at nesting depth 2 and must still refresh the bf16 shadow on every outermost call (default policy), or
    after an optimizer step / in-place update (tracked policy): a second forward after a weight update must change."""
    from icka_amd.config import BertConfig
    from icka_amd.modeling import BertAttention
    cfg = BertConfig(512, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=256)
    att = BertAttention(cfg)
    synth.fill_module_(att)
    att = att.cuda().eval()
    x = (torch.randn(2, 16, 128, generator=torch.Generator().manual_seed(0)) * 0.5).cuda().to(torch.bfloat16)
    mask = torch.zeros(2, 1, 1, 16, device="cuda")
    y0 = att(x, mask).float().clone()
    y0b = att(x, mask).float().clone()
    assert torch.equal(y0, y0b)
    with torch.no_grad():
        for p in att.parameters():              # BertAdam-style update through .data: no version counter moves
            p.data.mul_(1.25)
    y1 = att(x, mask).float().clone()
    A = att._icka_arena
    assert _shadow_ok(A), "stale bf16 weights after a .data update of a standalone BertAttention"
    assert (y1 - y0).abs().max().item() > 1e-3
    # tracked policy: an in-place update that bumps the version counters is seen at depth 2 as well
    A.shadow_policy = "tracked"
    att(x, mask)
    with torch.no_grad():
        att.output.dense.weight.mul_(0.5)
    y2 = att(x, mask).float().clone()
    assert _shadow_ok(A)
    assert (y2 - y1).abs().max().item() > 1e-3
    opt = torch.optim.SGD(att.parameters(), lr=0.5)
    att(x, mask).float().sum().backward()
    opt.step()                                   # seen through the global post-step hook
    att(x, mask)
    assert _shadow_ok(A)


def test_one_shadow_cast_per_outermost_forward(monkeypatch):
    """the default policy used to re-cast the arena 2-3 times per model forward (trunk + BertModel.encode +
    scalar gate); it is ONE cast per contiguous range and outermost call now."""
    from icka_amd import kernels as K
    model, args, labels = _model()
    model(*args, labels=labels)                  # builds the arena
    A = model._icka_arena
    calls = []
    real = K.cast_f32_to_bf16
    monkeypatch.setattr(K, "cast_f32_to_bf16", lambda s, d: (calls.append(s.numel()), real(s, d))[1])
    model(*args, labels=labels)
    assert len(calls) == len(A._cast_ranges), (len(calls), len(A._cast_ranges))
