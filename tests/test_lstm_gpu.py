"""GPU: the BiLSTM layer (icka_lstm_* + GEMM kernels) against torch.nn.LSTM evaluated on the CPU in fp32 with the same
weights (ATen is the arithmetic the reference itself calls: Cross_Modal_Interaction_Module.py:905-908, :1042)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TAGGER_GRAD_BAR = 1.6e-2   # 2x the 7.8e-3 measured on MI355X (printed by the test)


@pytest.mark.parametrize("persistent", [1, 2, 0])
@pytest.mark.parametrize("B,S,H", [(2, 5, 32), (4, 16, 64), (32, 128, 768), (3, 40, 256), (40, 9, 64), (32, 24, 1024), (7, 12, 1024),
                                   (20, 33, 512)])
def test_bilstm_forward_backward_against_aten(B, S, H, persistent):
    from icka_amd import _lib
    from icka_amd.lstm import BiLSTM
    lib = _lib.load()
    torch.manual_seed(B * 100 + S)
    ref = torch.nn.LSTM(H, H, batch_first=True, bidirectional=True)
    mine = BiLSTM(H, H)
    # 1: one persistent launch, forward hand-off by tagged data words (H % 256 == 0) / 2: the same with step tickets in
    # both directions / 0: one launch per step -- a per-call argument of icka_lstm_fwd / _bwd (no process-wide switch)
    mine.recurrence_flags = {1: 0, 2: _lib.LSTM_TICKETS, 0: _lib.LSTM_PER_STEP}[persistent]
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
    torch.cuda.synchronize()
    assert lib.icka_lstm_barrier_error() == 0
    print("\n[BiLSTM B%d S%d H%d persistent=%d] max abs out err %.3e, dx rel %.3e"
          % (B, S, H, persistent, err, rel(xg.grad, xr.grad)))


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


def test_gate_1_tagger_end_to_end_against_oracle_composition():
    """trunk -> BiLSTM -> classifier -> CRF (Cross_Modal_Interaction_Module.py:2383-2483, the `_gate_1` class): emissions,
    token_mean CRF loss, gradients and decoded tags against the CPU composition of the trunk oracle, ATen's nn.LSTM and
    the CRF oracle on the same by-key weights."""
    import torch.nn.functional as F
    from icka_amd import synth
    from icka_amd.config import BertConfig
    from icka_amd.modeling import MTCCMBertForMMTokenClassificationCRF_gate_1
    from oracle import crf_oracle as OC
    from oracle import mner_oracle as O
    B, S, H, C = 4, 32, 128, 13
    cfg = BertConfig(512, hidden_size=H, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                     max_position_embeddings=64)
    model = MTCCMBertForMMTokenClassificationCRF_gate_1(cfg, num_labels=C)
    synth.fill_module_(model)
    with torch.no_grad():   # LSTM weights at their usual scale (by-key N(0, 0.02) would leave the gates linear)
        for n, p in model.lstm.named_parameters():
            p.mul_(4.0)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    b = synth.synthetic_batch(B, S, 49, vocab_size=512, seed=3, layout="BCHW")
    g = {k: v.cuda() for k, v in b.items()}
    args = dict(input_ids=g["input_ids"], segment_ids=g["segment_ids"], input_mask=g["input_mask"],
                ori_input_ids=g["input_ids"], ori_input_mask=g["input_mask"], ori_segment_ids=g["segment_ids"],
                added_attention_mask=g["added_attention_mask"], visual_embeds_att=g["visual_embeds_att"],
                output_mask=g["input_mask"], labels=g["labels"])
    em = model(**args)                       # mode=None -> emissions
    loss = model(mode="train", **args)
    loss.backward()
    pred, dev_loss = model(mode="dev", **args)
    assert abs(dev_loss.item() - loss.item()) < 1e-4
    assert model(mode="test", **args) == pred
    # ---- CPU composition
    ocfg = O.OracleConfig(vocab_size=512, hidden_size=H, num_hidden_layers=2, num_attention_heads=2,
                          intermediate_size=256, max_position_embeddings=64)
    _, cross, _ = O.mner_trunk(P, ocfg, b["input_ids"], b["segment_ids"], b["input_mask"], b["added_attention_mask"],
                            b["visual_embeds_att"], 1, 49, False)
    lstm = torch.nn.LSTM(H, H, batch_first=True, bidirectional=True)
    lstm_sd = {k[len("lstm."):]: v for k, v in P.items() if k.startswith("lstm.")}
    x, _ = torch.func.functional_call(lstm, lstm_sd, (cross,))
    ref_em = F.linear(x, P["classifier.weight"], P["classifier.bias"])
    mask = b["input_mask"].bool()
    crfP = [P["crf.start_transitions"], P["crf.end_transitions"], P["crf.transitions"]]
    rloss = -OC.crf_reduce(OC.crf_llh(ref_em, b["labels"], mask, *crfP), mask, "token_mean")
    rloss.backward()
    err = (em.float().cpu() - ref_em.detach()).abs().max().item()
    assert err < 2e-2, err
    assert abs(loss.item() - rloss.item()) < 2e-2 * max(1.0, abs(rloss.item()))
    gmax = max(v.grad.norm().item() for v in P.values() if v.grad is not None)
    worst = 0.0
    for k, p in model.named_parameters():
        if P[k].grad is None:
            continue
        worst = max(worst, ((p.grad.float().cpu() - P[k].grad).norm() / (P[k].grad.norm() + 1e-4 * gmax)).item())
    ref_pred = OC.crf_decode(em.float().cpu(), mask, *[p.detach() for p in crfP])
    assert pred == ref_pred
    print("\n[_gate_1 tagger] emissions max abs err %.3e, loss %.4f (oracle %.4f), worst grad rel err %.3e"
          % (err, loss.item(), rloss.item(), worst))
    assert worst < TAGGER_GRAD_BAR, worst


@pytest.mark.parametrize("handoff", [1, 0])
@pytest.mark.parametrize("phase", ["forward", "backward"])
def test_a_failed_handoff_is_never_silent(handoff, phase, request):
    """VERDICT r02 #4 / ADVICE: a hand-off wait of a persistent launch that gives up must not pass for a result.  The test
    hook makes block (0, direction 0) skip publishing step 3 (with a small poll budget, so the time-out takes
    milliseconds): the outputs of that call must be NaN-poisoned and the NEXT host touch-point must raise -- BiLSTM.forward
    itself and a GraphedStep replay -- after which the error is cleared and the layer works again.
    (reference: nn.LSTM at Cross_Modal_Interaction_Module.py:905-908, :1042 cannot fail this way.)"""
    from icka_amd import _lib
    from icka_amd import kernels as K
    from icka_amd.lstm import BiLSTM
    lib = _lib.load()
    form = 0 if handoff == 1 else _lib.LSTM_TICKETS

    def restore():
        lib.icka_lstm_test_hooks(0, -1)
        lib.icka_lstm_clear_error()
    request.addfinalizer(restore)
    B, S, H = 8, 12, 256
    torch.manual_seed(5)
    m = BiLSTM(H, H).cuda()
    m.recurrence_flags = form
    x = (torch.randn(B, S, H) * 0.5).cuda().requires_grad_(True)
    out, _ = m(x)                      # healthy call first (also maps the host-visible error word)
    out.float().sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(x.grad.float()).all()
    assert lib.icka_lstm_barrier_error() == 0
    x.grad = None
    m.zero_grad()
    if phase == "forward":
        lib.icka_lstm_test_hooks(2000, 3)
        out, _ = m(x)
        lib.icka_lstm_test_hooks(0, -1)
        torch.cuda.synchronize()
        assert torch.isnan(out.float()).any(), "a failed forward hand-off left finite outputs"
        # direction 0 is poisoned from step 4 on, for every batch row and unit
        assert torch.isnan(out.float()[:, 4:, :H]).all()
    else:
        out, _ = m(x)
        lib.icka_lstm_test_hooks(2000, 3)
        out.float().sum().backward()
        lib.icka_lstm_test_hooks(0, -1)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        assert torch.isnan(x.grad.float()).any(), "a failed backward hand-off left finite gradients"
        assert torch.isnan(m.weight_hh_l0.grad).any()
    assert lib.icka_lstm_barrier_error() == 1
    with pytest.raises(K.LstmHandoffError):
        m(x)                                   # next host touch-point raises ...
    assert lib.icka_lstm_barrier_error() == 0  # ... and clears the word
    out2, _ = m(x)
    torch.cuda.synchronize()
    assert torch.isfinite(out2.float()).all()
    print("\n[BiLSTM give-up path, handoff=%d, %s] outputs NaN-poisoned, error raised at the next touch-point" % (handoff, phase))


def test_a_failed_handoff_inside_a_graph_replay_is_raised_by_the_next_replay(request):
    from icka_amd import _lib
    from icka_amd import kernels as K
    from icka_amd.graph import GraphedStep
    from icka_amd.lstm import BiLSTM
    lib = _lib.load()
    request.addfinalizer(lambda: (lib.icka_lstm_test_hooks(0, -1), lib.icka_lstm_clear_error()))
    B, S, H = 8, 12, 256
    torch.manual_seed(6)
    m = BiLSTM(H, H).cuda()
    x = (torch.randn(B, S, H) * 0.5).cuda()

    def step():
        # the hook values are kernel arguments: set only while the step is being captured, they are baked into the graph's
        # launches (the eager warm-up steps stay healthy)
        cap = torch.cuda.is_current_stream_capturing()
        lib.icka_lstm_test_hooks(2000 if cap else 0, 3 if cap else -1)
        out, _ = m(x)
        loss = out.float().sum()
        loss.backward()
        return loss
    step()                                       # builds the arena, maps the error word
    torch.cuda.synchronize()
    g = GraphedStep(m, step, warmup=1)
    lib.icka_lstm_test_hooks(0, -1)
    torch.cuda.synchronize()
    assert lib.icka_lstm_barrier_error() == 0    # nothing has run with the hook yet (capture launches nothing)
    loss = g()
    torch.cuda.synchronize()
    assert torch.isnan(loss).item()
    with pytest.raises(K.LstmHandoffError):
        g()
    g.close()


def test_reserved_cus_take_the_per_step_launches_when_the_grid_no_longer_fits(request):
    """dp.GradReducer reserves CUs for RCCL's workgroups: a persistent grid that would need them falls back to one launch
    per step (same results)."""
    from icka_amd import _lib
    from icka_amd.lstm import BiLSTM
    lib = _lib.load()
    request.addfinalizer(lambda: lib.icka_lstm_set_reserved_cus(0))
    B, S, H = 20, 9, 512
    torch.manual_seed(7)
    m = BiLSTM(H, H).cuda()
    x = (torch.randn(B, S, H) * 0.5).cuda()
    with torch.no_grad():
        ref, _ = m(x)
        lib.icka_lstm_set_reserved_cus(torch.cuda.get_device_properties(0).multi_processor_count - 8)
        out, _ = m(x)
    torch.cuda.synchronize()
    assert (out.float() - ref.float()).abs().max().item() < 1e-2
    assert lib.icka_lstm_barrier_error() == 0


def test_persistent_launches_on_two_streams_take_turns():
    """The tagged-word / ticket buffers of the persistent recurrences are process-global (csrc/lstm.hip: LstmTurn): launches
    from two streams must be serialised by the library, not overlap and read each other's words."""
    from icka_amd import _lib
    from icka_amd.lstm import BiLSTM
    lib = _lib.load()
    torch.manual_seed(5)
    B, S, H = 32, 96, 768
    mods = [BiLSTM(H, H).cuda() for _ in range(2)]
    xs = [(torch.randn(B, S, H, device="cuda") * 0.5) for _ in range(2)]
    want = []
    for m, x in zip(mods, xs):                       # one after the other on the current stream
        xx = x.clone().requires_grad_(True)
        out, _ = m(xx)
        out.float().sum().backward()
        want.append((out.detach().clone(), xx.grad.clone()))
        m.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    got = [None, None]
    for rep in range(3):
        xg = [x.clone().requires_grad_(True) for x in xs]
        for i in (0, 1):
            streams[i].wait_stream(torch.cuda.current_stream())
        outs = [None, None]
        for i in (0, 1):                             # both forwards enqueued before either backward
            with torch.cuda.stream(streams[i]):
                outs[i] = mods[i](xg[i])[0]
        for i in (1, 0):
            with torch.cuda.stream(streams[i]):
                outs[i].float().sum().backward()
                got[i] = (outs[i].detach(), xg[i].grad)
        torch.cuda.synchronize()
        assert lib.icka_lstm_barrier_error() == 0
        for i in (0, 1):
            assert torch.equal(got[i][0], want[i][0]), (rep, i)
            assert torch.equal(got[i][1], want[i][1]), (rep, i)
            mods[i].zero_grad(set_to_none=True)
