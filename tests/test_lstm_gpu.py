"""GPU: the BiLSTM layer (icka_lstm_* + GEMM kernels) against torch.nn.LSTM evaluated on the CPU in fp32 with the same
weights (ATen is the arithmetic the reference itself calls: Cross_Modal_Interaction_Module.py:905-908, :1042)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,S,H", [(2, 5, 32), (4, 16, 64), (32, 128, 768), (3, 40, 256)])
def test_bilstm_forward_backward_against_aten(B, S, H):
    from icka_amd.lstm import BiLSTM
    torch.manual_seed(B * 100 + S)
    ref = torch.nn.LSTM(H, H, batch_first=True, bidirectional=True)
    mine = BiLSTM(H, H)
    mine.load_state_dict(ref.state_dict())          # same parameter names as nn.LSTM
    mine = mine.cuda()
    x = torch.randn(B, S, H) * 0.5
    xg = x.cuda().requires_grad_(True)
    out, (h_n, c_n) = mine(xg)
    xr = x.to(torch.bfloat16).float().requires_grad_(True)   # the kernels see the bf16-rounded input
    ro, (rh, rc) = ref(xr)
    assert out.shape == ro.shape and out.dtype == torch.bfloat16
    err = (out.float().cpu() - ro).abs().max().item()
    assert err < 2e-2, err
    assert (h_n.cpu() - rh).abs().max().item() < 2e-2 and (c_n.cpu() - rc).abs().max().item() < 3e-2
    w = torch.randn(B, S, 2 * H, generator=torch.Generator().manual_seed(1))
    (out.float() * w.cuda()).sum().backward()
    (ro * w).sum().backward()
    rel = lambda a, b: ((a.float().cpu() - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(xg.grad, xr.grad) < 3e-2
    for n, p in mine.named_parameters():
        assert rel(p.grad, dict(ref.named_parameters())[n].grad) < 3e-2, n
    print("\n[BiLSTM B%d S%d H%d] max abs out err %.3e, dx rel %.3e" % (B, S, H, err, rel(xg.grad, xr.grad)))


def test_bilstm_rejects_unsupported_configurations():
    from icka_amd.lstm import BiLSTM
    with pytest.raises(ValueError):
        BiLSTM(64, 64, num_layers=2)
    with pytest.raises(ValueError):
        BiLSTM(64, 48)
    m = BiLSTM(64, 64).cuda()
    with pytest.raises(TypeError):
        m(torch.zeros(2, 3, 64))
    with pytest.raises(ValueError):
        m(torch.zeros(2, 3, 32, device="cuda"))
