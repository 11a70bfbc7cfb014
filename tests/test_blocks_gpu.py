"""GPU: the boundary below the layer level -- sub-module forwards composed the way the reference composes them
(Cross_Modal_Interaction_Module.py:438-442, :451-454, :633-636, :646-650) against the fused layer and the CPU oracle, and
the a_transformers-convention shim (icka_amd/hf_style.py; a_transformers/modeling_bert.py:454-534, :537-631)."""
import copy

import pytest
import torch

import icka_amd
from icka_amd import synth
from icka_amd.config import BertConfig

pytestmark = pytest.mark.gpu

CFG = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, max_position_embeddings=64)


def _inputs(B=3, S=32, R=49, H=128, seed=3):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, S, H, generator=g)
    s2 = torch.randn(B, R, H, generator=g)
    lens = torch.randint(S // 4, S + 1, (B,), generator=g)
    m = (torch.arange(S)[None, :] < lens[:, None]).float()
    ext = ((1.0 - m) * -10000.0)[:, None, None, :]
    ext2 = torch.zeros(B, 1, 1, R)
    ext2[0, ..., R - 5:] = -10000.0
    return x, s2, ext, ext2


def _rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-12)).item()


@pytest.mark.parametrize("precision,out_tol,grad_tol", [("bf16", 2e-2, 8e-2), ("fp32", 1e-5, 1e-4)])
def test_composed_sub_modules_equal_the_fused_layers_and_the_oracle(precision, out_tol, grad_tol):
    from icka_amd.modeling import BertCrossAttentionLayer, BertLayer
    from oracle import mner_oracle as O
    cfg = BertConfig(512, **CFG)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.l = BertLayer(cfg)
            self.c = BertCrossAttentionLayer(cfg)

    fused = Holder()
    synth.fill_module_(fused)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in fused.state_dict().items()}
    comp = copy.deepcopy(fused)
    fused = icka_amd.set_precision(fused.cuda().eval(), precision)
    comp = icka_amd.set_precision(comp.cuda().eval(), precision)
    x, s2, ext, ext2 = _inputs()

    def run_fused(m, xi, s2i):
        return m.c(m.l(xi, ext.cuda()), s2i, ext2.cuda())

    def run_composed(m, xi, s2i):
        # BertLayer.forward :438-442
        att = m.l.attention(xi, ext.cuda())                        # BertAttention.forward :451-454
        h = m.l.output(m.l.intermediate(att), att)
        # BertCrossAttentionLayer.forward :646-650
        catt = m.c.attention(h, s2i, ext2.cuda())                  # BertCrossAttention.forward :633-636
        return m.c.output(m.c.intermediate(catt), catt)

    res = {}
    for name, m, fn in (("fused", fused, run_fused), ("composed", comp, run_composed)):
        xi = x.cuda().requires_grad_(True)
        s2i = s2.cuda().requires_grad_(True)
        y = fn(m, xi, s2i)
        y.float().square().sum().backward()
        res[name] = (y.detach().float().cpu(), xi.grad.float().cpu(), s2i.grad.float().cpu(),
                     {k: p.grad.detach().float().cpu() for k, p in m.named_parameters()})
    # oracle
    ocfg = O.OracleConfig(vocab_size=512, **CFG)
    xr, sr = x.clone().requires_grad_(True), s2.clone().requires_grad_(True)
    yr = O.cross_layer(P, "c", O.bert_layer(P, "l", xr, ext, ocfg, False), sr, ext2, ocfg, False)
    yr.square().sum().backward()
    for name in ("fused", "composed"):
        y, dx, ds2, g = res[name]
        assert (y - yr.detach()).abs().max().item() < out_tol * max(1.0, yr.abs().max().item()), name
        assert _rel(dx, xr.grad) < grad_tol and _rel(ds2, sr.grad) < grad_tol, name
        worst = max(_rel(g[k], P[k].grad) for k in g if P[k].grad is not None and P[k].grad.norm() > 1e-6)
        print("\n[%s %s] out err %.2e, dx rel %.2e, worst param grad rel %.2e"
              % (precision, name, (y - yr.detach()).abs().max().item(), _rel(dx, xr.grad), worst))
        assert worst < 2 * grad_tol, (name, worst)
    # the composed path is the same arithmetic as the fused one (only the gradient fan-in order differs)
    assert (res["fused"][0] - res["composed"][0]).abs().max().item() < (1e-6 if precision == "fp32" else 4e-2)


def test_sub_module_forwards_standalone_shapes_and_dropout():
    from icka_amd.modeling import BertIntermediate, BertSelfAttention, BertSelfOutput
    cfg = BertConfig(512, **CFG)
    sa, so, it = BertSelfAttention(cfg).cuda(), BertSelfOutput(cfg).cuda(), BertIntermediate(cfg).cuda()
    x, _, ext, _ = _inputs()
    ctx = sa(x.cuda(), ext.cuda())
    assert tuple(ctx.shape) == (3, 32, 128) and ctx.dtype == torch.bfloat16
    y = so(ctx, x.cuda())
    assert tuple(y.shape) == (3, 32, 128)
    g = it(y)
    assert tuple(g.shape) == (3, 32, 256)
    so.train()
    y1, y2 = so(ctx, x.cuda()), so(ctx, x.cuda())
    assert not torch.equal(y1, y2)                                  # dropout active in train mode
    with pytest.raises(TypeError):
        sa(x, ext)                                                  # CPU tensors are refused


def test_hf_style_blocks_follow_the_a_transformers_convention():
    from icka_amd import hf_style as HF
    from icka_amd.modeling import BertEncoder
    cfg = BertConfig(512, **CFG)
    ours = BertEncoder(cfg)
    synth.fill_module_(ours)
    hf = HF.BertEncoder(cfg)
    hf.load_state_dict(ours.state_dict(), strict=True)             # same keys both ways
    ours.load_state_dict(hf.state_dict(), strict=True)
    ours, hf = ours.cuda().eval(), hf.cuda().eval()
    x, _, ext, _ = _inputs()
    ref = ours(x.cuda(), ext.cuda(), output_all_encoded_layers=True)
    out = hf(x.cuda(), attention_mask=ext.cuda(), head_mask=[None] * 2, output_hidden_states=True, return_dict=True)
    assert torch.equal(out.last_hidden_state, ref[-1]) and torch.equal(out[0], ref[-1])
    assert len(out.hidden_states) == 3 and torch.equal(out.hidden_states[1], ref[0])
    tup = hf(x.cuda(), attention_mask=ext.cuda(), return_dict=False)
    assert isinstance(tup, tuple) and len(tup) == 1 and torch.equal(tup[0], ref[-1])
    layer_out = hf.layer[0](x.cuda(), attention_mask=ext.cuda(), output_attentions=False)
    assert isinstance(layer_out, tuple) and len(layer_out) == 1 and torch.equal(layer_out[0], ref[0])
    att = HF.BertAttention(cfg)
    att.load_state_dict(hf.layer[0].attention.state_dict(), strict=True)
    a = att.cuda().eval()(x.cuda(), attention_mask=ext.cuda())
    assert isinstance(a, tuple) and tuple(a[0].shape) == (3, 32, 128)
    no_mask = hf.layer[0](x.cuda())[0]                              # attention_mask=None -> nothing masked
    assert torch.isfinite(no_mask.float()).all()
    for bad in (dict(output_attentions=True), dict(head_mask=torch.ones(2)), dict(encoder_hidden_states=x.cuda())):
        with pytest.raises(NotImplementedError):
            hf.layer[0](x.cuda(), attention_mask=ext.cuda(), **bad)
    emb = HF.BertEmbeddings(cfg)
    assert "position_ids" in emb.state_dict()
    ids = torch.randint(1, 512, (3, 32))
    e = emb.cuda().eval()(input_ids=ids.cuda(), token_type_ids=torch.zeros_like(ids).cuda())
    assert tuple(e.shape) == (3, 32, 128)
