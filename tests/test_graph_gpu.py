"""GPU: graph.GraphedStep(wgrad_stream=True) -- the grouped weight-gradient launches replayed on a side stream behind flag
waits instead of inside the captured graph -- gives the gradients of the plain captured step bit for bit (same kernels,
same operands, same order of the accumulations into each gradient; only the stream differs), replay after replay, and
keeps the GraphedStep gradient contract (reference loop: zero_grad after every step, My_cross_attention.py:843)."""
import pytest
import torch

from icka_amd import synth

pytestmark = pytest.mark.gpu


def _model_and_step(seed):
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF
    cfg = BertConfig(512, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024,
                     max_position_embeddings=128)
    model = MTCCMBertForMMTokenClassificationCRF(cfg, layer_num1=1, num_labels=13, regions=36)
    synth.fill_module_(model)
    model = model.cuda().eval()
    g = {k: v.cuda() for k, v in synth.synthetic_batch(8, 64, 36, vocab_size=512, seed=seed).items()}
    one = torch.ones((), device="cuda")

    def step():
        loss = model(g["input_ids"], g["segment_ids"], g["input_mask"], g["added_attention_mask"], g["visual_embeds_mean"],
                     g["visual_embeds_att"], labels=g["labels"])
        loss.backward(gradient=one)
        return loss
    return model, step


def test_wgrad_stream_step_equals_the_plain_captured_step():
    from icka_amd import _lib
    from icka_amd.graph import GraphedStep
    model, step = _model_and_step(3)
    model.zero_grad()
    step()
    plain = GraphedStep(model, step)
    plain()
    torch.cuda.synchronize()
    ref = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    ref_loss = plain.loss.item()
    plain.close()
    model.zero_grad()
    gs = GraphedStep(model, step, wgrad_stream=True)
    assert len(gs.wgrad.items) >= 5          # 3 BERT layers + cross layer + gated head
    for it in range(3):
        loss = gs()
        torch.cuda.synchronize()
        assert loss.item() == ref_loss
        for n, p in model.named_parameters():
            if n in ref:
                assert p.grad is not None, n
                if "embeddings.word" in n or "embeddings.position" in n:
                    # f32 atomics into the table rows: the order of the adds (repeated ids) is not fixed, in either form
                    assert (p.grad - ref[n]).abs().max().item() < 1e-6 * ref[n].abs().max().item(), n
                else:
                    assert torch.equal(p.grad, ref[n]), (it, n, (p.grad - ref[n]).abs().max().item())
        model.zero_grad()
    assert int(gs.sync[0].item()) == 3 and _lib.load().icka_dp_error() == 0
    flags = gs.sync[gs.WG_FLAG0:gs.WG_FLAG0 + len(gs.wgrad.items)].tolist()
    assert flags == [3] * len(gs.wgrad.items)
    gs.close()
