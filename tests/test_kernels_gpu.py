"""GPU: every C-ABI kernel against a plain PyTorch fp32 evaluation of the same op on the same (bf16-rounded) inputs.
The end-to-end parity against the CPU oracle / golden fixtures lives in test_model_gpu.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def _k():
    from icka_amd import kernels
    return kernels


def rel_err(a, b):
    a, b = a.float(), b.float()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def rnd(*shape, scale=1.0, seed=0, dtype=BF16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


# --------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (4096, 768, 768), (100, 13, 1536), (77, 200, 72),
                                   (64, 40, 13), (1152, 768, 2048), (256, 64, 192), (1024, 64, 576), (128, 64, 64)])
def test_gemm_nt(M, N, K):
    k = _k()
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias = rnd(N, seed=3, dtype=F32)
    out = torch.empty(M, N, dtype=BF16, device="cuda")
    k.gemm(k.GEMM_NT, A, B, out, bias=bias)
    ref = A.float() @ B.float().t() + bias
    assert rel_err(out, ref) < 1e-2
    outf = torch.empty(M, N, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NT, A, B, outf)
    assert rel_err(outf, A.float() @ B.float().t()) < 1e-4


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (4096, 768, 768), (4096, 768, 3072), (4096, 2304, 768), (4096, 3072, 768),
                                   (8192, 1024, 1024), (100, 13, 1536), (77, 200, 72), (1152, 768, 2048), (1024, 64, 576)])
def test_gemm_nt_fp16_operands(M, N, K):
    """the "mixed16" forward GEMMs: fp16 operands on v_mfma_f32_16x16x32_f16, outputs bf16 / f32 / fp16 (+ bf16 copy);
    every NT kernel of the dispatch (256x192, 128x96, 128x128, two-blocks-per-CU, 128x64, general path).  The fp16 product
    must sit ~8x closer to the f32 product than the bf16 one (11 vs 8 significand bits)."""
    k = _k()
    F16 = torch.float16
    A32, B32 = rnd(M, K, seed=1, dtype=F32), rnd(N, K, seed=2, dtype=F32)
    A, B = A32.to(F16), B32.to(F16)
    bias = rnd(N, seed=3, dtype=F32)
    ref = A.float() @ B.float().t()
    outf = torch.empty(M, N, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NT, A, B, outf)
    assert rel_err(outf, ref) < 1e-4
    exact = A32 @ B32.t()
    outb = torch.empty(M, N, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NT, A32.to(BF16), B32.to(BF16), outb)
    e16, eb = (outf - exact).norm().item(), (outb - exact).norm().item()
    assert e16 < 0.25 * eb, (e16, eb)
    out = torch.empty(M, N, dtype=BF16, device="cuda")
    k.gemm(k.GEMM_NT, A, B, out, bias=bias)
    assert rel_err(out, ref + bias) < 1e-2
    oh, ob = torch.empty(M, N, dtype=F16, device="cuda"), torch.empty(M, N, dtype=BF16, device="cuda")
    k.gemm(k.GEMM_NT, A, B, oh, bias=bias, out3=ob)
    assert rel_err(oh, ref + bias) < 2e-3
    assert rel_err(ob, ref + bias) < 1e-2
    # GELU epilogue: z -> out2 (bf16), gelu(z) -> fp16 main output + bf16 copy
    z = torch.empty(M, N, dtype=BF16, device="cuda")
    k.gemm(k.GEMM_NT, A, B, oh, bias=bias, epilogue=k.EPI_GELU, out2=z, out3=ob)
    zr = (ref + bias) * 0.05
    A2 = (A32 * 0.05).to(F16)
    k.gemm(k.GEMM_NT, A2, B, oh, bias=bias * 0.05, epilogue=k.EPI_GELU, out2=z, out3=ob)
    zr = A2.float() @ B.float().t() + bias * 0.05
    gr = torch.nn.functional.gelu(zr)
    assert rel_err(z, zr) < 1e-2
    assert (oh.float() - gr).abs().max().item() < 2e-3 * max(1.0, gr.abs().max().item())
    assert (ob.float() - gr).abs().max().item() < 1e-2 * max(1.0, gr.abs().max().item())
    with pytest.raises(ValueError):
        k.gemm(k.GEMM_NN, A, B.t().contiguous(), outf)          # fp16 operands are NT only
    with pytest.raises(TypeError):
        k.gemm(k.GEMM_NT, A, B32.to(BF16), outf)                # mixed operand types


def test_gemm_fp16_output_saturates():
    k = _k()
    F16 = torch.float16
    A = torch.full((128, 64), 200.0, dtype=F16, device="cuda")
    B = torch.full((128, 64), 200.0, dtype=F16, device="cuda")
    oh = torch.empty(128, 128, dtype=F16, device="cuda")
    k.gemm(k.GEMM_NT, A, B, oh)       # 64 * 4e4 = 2.56e6 > 65504
    assert torch.isfinite(oh).all() and (oh.float() == 65504.0).all()


def test_gemm_identity_asymmetric():
    """A = I with an asymmetric B catches a transposed / permuted C write (cdna guide section 3)."""
    k = _k()
    n = 128
    A = torch.eye(n, dtype=BF16, device="cuda")
    B = (torch.arange(n * n, device="cuda").reshape(n, n) % 251).to(BF16)
    out = torch.empty(n, n, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NT, A, B, out)
    assert torch.equal(out, B.float().t())
    k.gemm(k.GEMM_NN, A, B, out)
    assert torch.equal(out, B.float())
    k.gemm(k.GEMM_TN, A, B, out)
    assert torch.equal(out, B.float())
    k.gemm(k.GEMM_TN, B, A, out)
    assert torch.equal(out, B.float().t())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (4096, 768, 2304), (100, 1536, 13), (200, 72, 77), (256, 3072, 768)])
def test_gemm_nn(M, N, K):
    k = _k()
    A, B = rnd(M, K, seed=1), rnd(K, N, seed=2)
    if K % 8:   # k-contiguous operand rows must be 16-byte aligned: pad the leading dimension like the callers do
        Kp = (K + 7) // 8 * 8
        Ap = torch.zeros(M, Kp, dtype=BF16, device="cuda")
        Ap[:, :K] = A
        A = Ap[:, :K]
    out = torch.empty(M, N, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NN, A, B, out)
    assert rel_err(out, A.float() @ B.float()) < 1e-4


@pytest.mark.parametrize("Kt,M,N", [(64, 128, 128), (4096, 768, 768), (4096, 2304, 768), (300, 13, 1536), (1000, 72, 200),
                                    (4096, 768, 3072), (4096, 13, 768), (2048, 13, 1536)])
def test_gemm_tn(Kt, M, N):
    k = _k()
    A, B = rnd(Kt, M if M % 8 == 0 else 16, seed=1), rnd(Kt, N, seed=2)
    A = A[:, :M]
    out = torch.full((M, N), 1.0, dtype=F32, device="cuda")
    k.gemm(k.GEMM_TN, A, B, out, beta=1.0)   # accumulate into existing gradient
    ref = A.float().t() @ B.float() + 1.0
    assert rel_err(out, ref) < 1e-4


def test_gemm_epilogues_and_dual_k():
    k = _k()
    M, N, K = 256, 384, 128
    A, B = rnd(M, K, seed=1, scale=0.5), rnd(N, K, seed=2, scale=0.5)
    bias = rnd(N, seed=3, dtype=F32)
    acc = A.float() @ B.float().t() + bias
    # GELU (two outputs)
    g = torch.empty(M, N, dtype=BF16, device="cuda"); z = torch.empty_like(g)
    k.gemm(k.GEMM_NT, A, B, g, bias=bias, epilogue=k.EPI_GELU, out2=z)
    assert rel_err(z, acc) < 1e-2
    assert rel_err(g, torch.nn.functional.gelu(acc)) < 1e-2
    # DGELU
    aux = rnd(M, N, seed=4)
    o = torch.empty(M, N, dtype=BF16, device="cuda")
    k.gemm(k.GEMM_NT, A, B, o, epilogue=k.EPI_DGELU, aux=aux)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel_err(o, (A.float() @ B.float().t()) * x.grad) < 1e-2
    # ADD
    k.gemm(k.GEMM_NT, A, B, o, bias=bias, epilogue=k.EPI_ADD, aux=aux)
    assert rel_err(o, acc + aux.float()) < 1e-2
    # GATE
    gate = torch.empty_like(o)
    k.gemm(k.GEMM_NT, A, B, o, bias=bias, epilogue=k.EPI_GATE, aux=aux, out2=gate)
    assert rel_err(gate, torch.sigmoid(acc)) < 1e-2
    assert rel_err(o, torch.sigmoid(acc) * aux.float()) < 1e-2
    # TANH
    k.gemm(k.GEMM_NT, A, B, o, bias=bias, epilogue=k.EPI_TANH)
    assert rel_err(o, torch.tanh(acc)) < 1e-2
    # dual K segment (cat along K without materialising it); A halves live in a wider buffer (strided views)
    wide = rnd(M, 4 * K, seed=5, scale=0.5)
    A1, A2 = wide[:, :K], wide[:, 2 * K:3 * K]
    B1, B2 = rnd(N, K, seed=6, scale=0.5), rnd(N, K, seed=7, scale=0.5)
    of = torch.empty(M, N, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NT, A1, B1, of, A2=A2, B2=B2)
    ref = A1.float() @ B1.float().t() + A2.float() @ B2.float().t()
    assert rel_err(of, ref) < 1e-4
    # bf16 accumulate (beta) and alpha
    o.fill_(1.0)
    k.gemm(k.GEMM_NT, A, B, o, alpha=0.5, beta=1.0)
    assert rel_err(o, 0.5 * (A.float() @ B.float().t()) + 1.0) < 1e-2


@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("op", ["NT", "NN"])
@pytest.mark.parametrize("M,N,K", [(4096, 2304, 768), (4096, 3072, 768), (2048, 3072, 64), (4096, 1536, 1024), (4096, 3072, 192),
                                   (8192, 1024, 1024), (8192, 1024, 3072), (8192, 768, 768), (6144, 1024, 192)])
def test_gemm_wide_tiles(M, N, K, op, wide):
    """256x192-tile kernel (qkv / ffn-up / d-ffn-down shapes: 128..256 tiles of one round) and its 256x128-tile form (192..256
    tiles: N = 1024 / 768 at M = 8192, long and short reductions) against fp32 matmul, with
    every epilogue those GEMMs use: bias, GELU (two outputs), GELU' (aux), fan-in add, f32 and bf16 outputs, odd and even
    k-tile counts; wide=0 runs the same calls on the 128x128 path."""
    k = _k()
    T = k.gemm_tune(wide_tiles=bool(wide))      # a per-call word of the descriptor: no process-wide switch
    A = rnd(M, K, seed=1, scale=0.5)
    B = rnd(N, K, seed=2, scale=0.5) if op == "NT" else rnd(K, N, seed=2, scale=0.5)
    kop = k.GEMM_NT if op == "NT" else k.GEMM_NN
    bias = rnd(N, seed=3, dtype=F32)
    ref = A.float() @ (B.float().t() if op == "NT" else B.float())
    o = torch.empty(M, N, dtype=BF16, device="cuda")
    k.gemm(kop, A, B, o, tune=T)
    assert rel_err(o, ref) < 1e-2
    of = torch.empty(M, N, dtype=F32, device="cuda")
    k.gemm(kop, A, B, of, bias=bias, tune=T)
    assert rel_err(of, ref + bias) < 1e-4
    z = torch.empty_like(o)
    k.gemm(kop, A, B, o, bias=bias, epilogue=k.EPI_GELU, out2=z, tune=T)
    assert rel_err(z, ref + bias) < 1e-2 and rel_err(o, torch.nn.functional.gelu(ref + bias)) < 1e-2
    aux = rnd(M, N, seed=4)
    k.gemm(kop, A, B, o, epilogue=k.EPI_DGELU, aux=aux, tune=T)
    x = aux.float().requires_grad_(True)
    torch.nn.functional.gelu(x).sum().backward()
    assert rel_err(o, ref * x.grad) < 1e-2
    k.gemm(kop, A, B, o, epilogue=k.EPI_ADD, aux=aux, tune=T)
    assert rel_err(o, ref + aux.float()) < 1e-2
    ov = torch.empty(M, N + 64, dtype=BF16, device="cuda")[:, :N]      # strided output view
    k.gemm(kop, A, B, ov, alpha=0.5, tune=T)
    assert rel_err(ov, 0.5 * ref) < 1e-2


@pytest.mark.parametrize("M,N,K,op", [(4096, 3072, 768, "NT"), (4096, 2304, 768, "NT"), (4096, 3072, 768, "NN"), (2048, 3072, 192, "NT"),
                                      (8192, 1024, 1024, "NN"), (6144, 1024, 192, "NT")])
def test_gemm_wide_tiles_xcd_cut_does_not_change_results(M, N, K, op):
    """The 12-wave kernel cuts its tile grid over the 8 XCDs per shape (ICKA_TUNE_W3_GRID: default = pick the cut, 8 / 4 / 2 / 1 =
    force the number of patch rows where it divides the grid): every forced cut covers every tile exactly once -- bitwise the
    result of the default."""
    k = _k()
    A = rnd(M, K, seed=11, scale=0.5)
    B = rnd(N, K, seed=12, scale=0.5) if op == "NT" else rnd(K, N, seed=12, scale=0.5)
    kop = k.GEMM_NT if op == "NT" else k.GEMM_NN
    ref = A.float() @ (B.float().t() if op == "NT" else B.float())
    outs = []
    for pm in (0, 8, 4, 2, 1):
        o = torch.full((M, N), float("nan"), dtype=BF16, device="cuda")
        k.gemm(kop, A, B, o, tune=k.gemm_tune(w3_grid=pm) if pm else 0)
        assert torch.isfinite(o.float()).all(), pm          # no tile left out
        outs.append(o)
    assert rel_err(outs[0], ref) < 1e-2
    for pm, o in zip((8, 4, 2, 1), outs[1:]):
        assert torch.equal(o, outs[0]), pm


@pytest.mark.parametrize("tile_n", [96, 128])
@pytest.mark.parametrize("M,N,K", [(128, 384, 64), (4096, 768, 768), (512, 2304, 768), (256, 768, 3072)])
def test_gemm_tile_widths(M, N, K, tile_n):
    """128x96 and 128x128 output tiles of the warp-specialised path, all three layouts, fused epilogues."""
    k = _k()
    T = k.gemm_tune(tile_n=tile_n)
    A, Bt, Bn = rnd(M, K, seed=1, scale=0.5), rnd(N, K, seed=2, scale=0.5), rnd(K, N, seed=3, scale=0.5)
    bias, aux = rnd(N, seed=4, dtype=F32), rnd(M, N, seed=5)
    acc = A.float() @ Bt.float().t() + bias
    g = torch.empty(M, N, dtype=BF16, device="cuda"); z = torch.empty_like(g)
    k.gemm(k.GEMM_NT, A, Bt, g, bias=bias, epilogue=k.EPI_GELU, out2=z, tune=T)
    assert rel_err(z, acc) < 1e-2 and rel_err(g, torch.nn.functional.gelu(acc)) < 1e-2
    of = torch.full((M, N), 1.0, dtype=F32, device="cuda")
    k.gemm(k.GEMM_NT, A, Bt, of, beta=1.0, tune=T)
    assert rel_err(of, A.float() @ Bt.float().t() + 1.0) < 1e-4
    o = torch.empty(M, N, dtype=BF16, device="cuda")
    for direct in (1, 0):   # plain outputs: straight from the accumulators, or through the LDS C tile
        T = k.gemm_tune(tile_n=tile_n, direct_epilogue=bool(direct))
        k.gemm(k.GEMM_NT, A, Bt, o, bias=bias, alpha=0.5, tune=T)
        assert rel_err(o, 0.5 * (A.float() @ Bt.float().t()) + bias) < 1e-2
        k.gemm(k.GEMM_NT, A, Bt, of, bias=bias, tune=T)
        assert rel_err(of, acc) < 1e-4
    T = k.gemm_tune(tile_n=tile_n)
    k.gemm(k.GEMM_NN, A, Bn, o, epilogue=k.EPI_ADD, aux=aux, tune=T)
    assert rel_err(o, A.float() @ Bn.float() + aux.float()) < 1e-2
    k.gemm(k.GEMM_NN, A, Bn, of, tune=T)
    assert rel_err(of, A.float() @ Bn.float()) < 1e-4
    # TN: dW[M2, N] = At^T . X  with the fused column sums (bias gradient) of At
    At, X = rnd(K, 256, seed=6), rnd(K, N, seed=7)
    dw = torch.full((256, N), 0.5, dtype=F32, device="cuda")
    cs = torch.zeros(256, dtype=F32, device="cuda")
    k.gemm(k.GEMM_TN, At, X, dw, beta=1.0, colsum_out=cs, tune=T)
    assert rel_err(dw, At.float().t() @ X.float() + 0.5) < 1e-4
    assert rel_err(cs, At.float().sum(0)) < 1e-4


def test_gemm_grouped_matches_individual_launches():
    """The four weight-gradient GEMMs of a layer in one launch (+ a ragged one that falls back)."""
    k = _k()
    T, H, I = 1024, 256, 512
    shapes = [(H, I), (I, H), (H, H), (3 * H, H), (13, H)]
    descs, outs, refs, keep = [], [], [], []
    for i, (m, n) in enumerate(shapes):
        A = rnd(T, m if m % 8 == 0 else 16, seed=10 + i)[:, :m]
        B = rnd(T, n, seed=20 + i)
        out = torch.full((m, n), 0.5, dtype=F32, device="cuda")
        descs.append(k.gemm_desc(k.GEMM_TN, A, B, out, beta=1.0))
        keep.append((A, B)); outs.append(out)
        refs.append(A.float().t() @ B.float() + 0.5)
    # the first two also produce their bias gradient (column sums of the dY operand) inside the GEMM
    cs = [torch.full((H,), 2.0, dtype=F32, device="cuda"), torch.zeros(I, dtype=F32, device="cuda")]
    descs[0] = k.gemm_desc(k.GEMM_TN, keep[0][0], keep[0][1], outs[0], beta=1.0, colsum_out=cs[0], colsum_accumulate=True)
    descs[1] = k.gemm_desc(k.GEMM_TN, keep[1][0], keep[1][1], outs[1], beta=1.0, colsum_out=cs[1], colsum_accumulate=False)
    k.gemm_grouped(descs)
    for o, r in zip(outs, refs):
        assert rel_err(o, r) < 1e-4
    assert rel_err(cs[0], keep[0][0].float().sum(0) + 2.0) < 1e-4
    assert rel_err(cs[1], keep[1][0].float().sum(0)) < 1e-4
    single = torch.zeros(I, dtype=F32, device="cuda")
    o2 = torch.empty(I, H, dtype=F32, device="cuda")
    k.gemm(k.GEMM_TN, keep[1][0], keep[1][1], o2, colsum_out=single)
    assert rel_err(single, keep[1][0].float().sum(0)) < 1e-4 and rel_err(o2, keep[1][0].float().t() @ keep[1][1].float()) < 1e-4


@pytest.mark.parametrize("big", [1, 2, 0])
def test_gemm_grouped_weight_gradient_layer_shapes(big):
    """The four weight-gradient GEMMs of a BERT layer as one grouped launch, overwrite mode (beta = 0) with fused bias
    gradients: 256x128-tile kernel with column-sum blocks in the same grid (big=1) vs the 128x128 group kernel."""
    k = _k()
    from icka_amd import _lib
    lib = _lib.load()
    TUNE = k.gemm_tune(big_tiles=big)    # a grouped launch takes the tune word of its first problem
    T, H, I = 1024, 768, 3072
    shapes = [(3 * H, H), (H, H), (I, H), (H, I)]
    descs, outs, refs, css, keep = [], [], [], [], []
    for i, (m, n) in enumerate(shapes):
        A, B = rnd(T, m, seed=30 + i, scale=0.5), rnd(T, n, seed=40 + i, scale=0.5)
        out = torch.full((m, n), 7.0, dtype=F32, device="cuda")          # overwritten, or accumulated into (i == 3)
        cs = torch.full((m,), 3.0, dtype=F32, device="cuda") if i != 1 else None
        descs.append(k.gemm_desc(k.GEMM_TN, A, B, out, beta=1.0 if i == 3 else 0.0, colsum_out=cs,
                                 colsum_accumulate=(i == 2), tune=TUNE))
        keep.append((A, B)); outs.append(out); css.append(cs)
        refs.append(A.float().t() @ B.float() + (7.0 if i == 3 else 0.0))
    # two LayerNorm slab reductions ride on the launch (dgamma / dbeta of the layer's two LayerNorms)
    slots = lib.icka_ln_slab_slots()
    reds, red_ref, red_out = [], [], []
    for r in range(2):
        nslab = 128 if r == 0 else 7
        part = rnd(nslab, slots, H, seed=60 + r, dtype=F32)
        o = [torch.full((H,), 5.0, dtype=F32, device="cuda") for _ in range(2)]
        reds.append(k.slab_reduction(part, nslab, H, o, accumulate=(r == 1)))
        red_ref.append([part[:, sl].sum(0) + (5.0 if r == 1 else 0.0) for sl in range(2)])
        red_out.append(o)
    k.gemm_grouped(descs, reductions=reds)
    for o, r in zip(outs, refs):
        assert rel_err(o, r) < 1e-4
    for i, cs in enumerate(css):
        if cs is not None:
            ref = keep[i][0].float().sum(0) + (3.0 if i == 2 else 0.0)
            assert rel_err(cs, ref) < 1e-4, i
    for o, r in zip(red_out, red_ref):
        assert rel_err(o[0], r[0]) < 1e-5 and rel_err(o[1], r[1]) < 1e-5
    # reductions alone (no GEMM to ride on): reduced by their own launches
    alone = [torch.zeros(H, dtype=F32, device="cuda") for _ in range(2)]
    part = rnd(33, slots, H, seed=70, dtype=F32)
    k.gemm_grouped([], reductions=[k.slab_reduction(part, 33, H, alone, accumulate=False)])
    assert rel_err(alone[0], part[:, 0].sum(0)) < 1e-5 and rel_err(alone[1], part[:, 1].sum(0)) < 1e-5


def test_gemm_bad_args():
    k = _k()
    A, B = rnd(64, 64), rnd(64, 64)
    with pytest.raises(TypeError):
        k.gemm(k.GEMM_NT, A.cpu(), B, torch.empty(64, 64, dtype=BF16, device="cuda"))
    with pytest.raises(ValueError):
        k.gemm(k.GEMM_NT, A, rnd(64, 32), torch.empty(64, 64, dtype=BF16, device="cuda"))


# ----------------------------------------------------------------------------------------------- LayerNorm
def _ln_ref(x, bias, res, gamma, beta, mask, eps):
    s = (x + bias) * mask + res
    mu = s.mean(-1, keepdim=True)
    var = ((s - mu) ** 2).mean(-1, keepdim=True)
    xh = (s - mu) / torch.sqrt(var + eps)
    return gamma * xh + beta, xh


@pytest.mark.parametrize("M,H,p", [(64, 128, 0.0), (4096, 768, 0.0), (1000, 1024, 0.1), (33, 768, 0.1)])
def test_ln_fwd_bwd(M, H, p):
    k = _k()
    x, res = rnd(M, H, seed=1), rnd(M, H, seed=2)
    bias, gamma, beta = rnd(H, seed=3, dtype=F32), rnd(H, seed=4, dtype=F32) + 1.0, rnd(H, seed=5, dtype=F32)
    seed = 0x1234_5678_9abc
    mask = k.dropout_mask(M * H, p, seed, "cuda").view(M, H)
    if p > 0:
        keep = (mask > 0).float().mean().item()
        assert abs(keep - (1 - p)) < 0.01
        assert torch.all((mask == 0) | ((mask - 1 / (1 - p)).abs() < 1e-6))
    y = torch.empty(M, H, dtype=BF16, device="cuda"); y2 = torch.empty(M, 2 * H, dtype=BF16, device="cuda")
    xhat = torch.empty_like(y); rstd = torch.empty(M, dtype=F32, device="cuda")
    yf = torch.empty(M, H, dtype=F32, device="cuda")
    k.ln_fwd(x, bias, res, gamma, beta, y, y2=y2[:, H:], y_f32=yf, xhat=xhat, rstd=rstd, eps=1e-12, p_drop=p, seed=seed)
    assert torch.equal(yf.to(BF16), y)
    ya = torch.empty_like(y)   # f32 input / f32 residual variants agree with the bf16 ones on the same values
    k.ln_fwd(x.float(), bias, res.float(), gamma, beta, ya, eps=1e-12, p_drop=p, seed=seed)
    assert torch.equal(ya, y)
    xf = x.float().requires_grad_(True); rf = res.float().requires_grad_(True)
    bf = bias.clone().requires_grad_(True); gf = gamma.clone().requires_grad_(True); b2 = beta.clone().requires_grad_(True)
    yref, xhref = _ln_ref(xf, bf, rf, gf, b2, mask, 1e-12)
    assert rel_err(y, yref) < 1e-2
    assert torch.equal(y2[:, H:], y)
    assert rel_err(xhat, xhref) < 1e-2
    dy = rnd(M, H, seed=6)
    yref.backward(dy.float())
    dres = torch.empty_like(y); dx = torch.empty_like(y)
    dg = torch.zeros(H, dtype=F32, device="cuda"); db = torch.zeros_like(dg); dbias = torch.zeros_like(dg)
    ws = k.ln_bwd_workspace(H, "cuda")
    k.ln_bwd(dy, xhat, rstd, gamma, dres=dres, dx=dx, dgamma=dg, dbeta=db, dbias=dbias, partials=ws, p_drop=p, seed=seed)
    assert rel_err(dres, rf.grad) < 2e-2
    assert rel_err(dx, xf.grad) < 2e-2
    assert rel_err(dg, gf.grad) < 2e-2
    assert rel_err(db, b2.grad) < 2e-2
    assert rel_err(dbias, bf.grad) < 2e-2
    # accumulate semantics (+=) and the dy2 fan-in
    k.ln_bwd(dy, xhat, rstd, gamma, dy2=dy, dres=dres, dgamma=dg, dbeta=db, partials=ws, p_drop=p, seed=seed)
    assert rel_err(db, 3 * b2.grad) < 2e-2
    assert rel_err(dg, 3 * gf.grad) < 2e-2
    assert rel_err(dres, 2 * rf.grad) < 2e-2
    # single-launch form (no dbias): overwrite mode, repeated launches (its device ticket re-arms itself), bitwise
    # reproducible column sums
    outs = []
    for _ in range(3):
        g2, b3 = torch.full_like(dg, 7.0), torch.full_like(db, 7.0)
        k.ln_bwd(dy, xhat, rstd, gamma, dres=dres, dx=dx, dgamma=g2, dbeta=b3, partials=ws, p_drop=p, seed=seed,
                 accumulate=False)
        outs.append((g2, b3))
    assert rel_err(outs[0][0], gf.grad) < 2e-2 and rel_err(outs[0][1], b2.grad) < 2e-2
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    assert rel_err(dx, xf.grad) < 2e-2


# ----------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k_, v, add_mask, dmask, B, h, Sq, Skv):
    """q [B*Sq, h*64] etc. (fp32, requires_grad) -> out [B*Sq, h*64]"""
    qh = q.view(B, Sq, h, 64).permute(0, 2, 1, 3)
    kh = k_.view(B, Skv, h, 64).permute(0, 2, 1, 3)
    vh = v.view(B, Skv, h, 64).permute(0, 2, 1, 3)
    s = qh @ kh.transpose(-1, -2) / 8.0 + add_mask[:, None, None, :]
    p = torch.softmax(s, -1) * dmask
    o = p @ vh
    return o.permute(0, 2, 1, 3).reshape(B * Sq, h * 64), torch.logsumexp(s, -1)


def test_attn_dropout_mask_matches_the_numpy_restatement():
    """icka_attn_dropout_mask (= what every attention kernel applies) against the numpy restatement of the hash in
    tests/test_dropout_hash_cpu.py: two decisions per 32-bit hash, hash(row * Skv + key / 2), even key -> low half."""
    import numpy as np
    from test_dropout_hash_cpu import keep_pair
    k = _k()
    k.set_dropout_nonce(None)   # (the restatement knows nothing of a graph's replay nonce)
    rows, Skv, p, seed = 37, 50, 0.1, 0x0123_4567_89ab_cdef
    m = k.attn_dropout_mask(rows, Skv, p, seed, "cuda").cpu().numpy()
    r, c = np.meshgrid(np.arange(rows, dtype=np.uint32), np.arange(Skv, dtype=np.uint32), indexing="ij")
    even, odd = keep_pair(seed, (r * np.uint32(Skv) + (c >> np.uint32(1))).astype(np.uint32), p)
    keep = np.where((c & 1) == 1, odd, even)
    assert np.array_equal(m > 0, keep)
    assert np.allclose(m[m > 0], 1.0 / (1.0 - p), rtol=1e-6)


@pytest.mark.parametrize("B,h,Sq,Skv,p", [(2, 2, 64, 64, 0.0), (2, 12, 128, 128, 0.0), (3, 2, 100, 49, 0.0),
                                          (2, 4, 128, 36, 0.1), (2, 3, 128, 128, 0.1), (1, 2, 32, 200, 0.0),
                                          (1, 16, 256, 256, 0.1), (2, 2, 49, 128, 0.1), (1, 2, 5, 7, 0.0),
                                          (2, 2, 128, 49, 0.1), (1, 3, 65, 65, 0.1), (1, 2, 300, 300, 0.1), (1, 2, 512, 512, 0.0),
                                          (2, 2, 200, 130, 0.1), (1, 2, 256, 64, 0.1), (1, 2, 320, 50, 0.0),
                                          (2, 3, 180, 180, 0.1), (1, 2, 192, 150, 0.0), (1, 2, 150, 192, 0.1)])
@pytest.mark.parametrize("whole_head", [True, False])
def test_attention_fwd_bwd(B, h, Sq, Skv, p, whole_head):
    k = _k()
    # whole_head=True: Sq, Skv <= 128 run the whole-head forward AND backward; up to 256 keys the forward keeps every score
    # of a 64-query block in registers (the backward then takes the tiled kernels); beyond that both are tiled
    tiled = not whole_head              # a per-call flag (ICKA_ATTN_TILED): no process-wide switch
    H = h * 64
    fused = rnd(B * Sq, 3 * H, seed=1)          # q lives inside a fused [M,3H] buffer (strided view)
    q = fused[:, H:2 * H]
    kk, v = rnd(B * Skv, H, seed=2), rnd(B * Skv, H, seed=3)
    lens = torch.randint(1, Skv + 1, (B,), generator=torch.Generator().manual_seed(4))
    m01 = (torch.arange(Skv)[None, :] < lens[:, None]).long().cuda()
    add_mask = torch.empty(B, Skv, dtype=F32, device="cuda")
    k.additive_mask(m01, Skv, add_mask)
    assert torch.equal(add_mask, (1.0 - m01.float()) * -10000.0)
    seed = 0xabcdef12345
    dmask = k.attn_dropout_mask(B * h * Sq, Skv, p, seed, "cuda").view(B, h, Sq, Skv)
    out = torch.empty(B * Sq, H, dtype=BF16, device="cuda")
    lse = torch.empty(B, h, Sq, dtype=F32, device="cuda")
    k.attn_fwd(q, kk, v, add_mask, out, lse, B, h, Sq, Skv, p_drop=p, seed=seed, tiled=tiled)
    qf, kf, vf = (t.float().contiguous().requires_grad_(True) for t in (q, kk, v))
    oref, lref = _attn_ref(qf, kf, vf, add_mask, dmask, B, h, Sq, Skv)
    assert rel_err(out, oref) < 1.5e-2
    assert (lse - lref).abs().max().item() < 2e-2
    dout = rnd(B * Sq, H, seed=5)
    oref.backward(dout.float())
    dq = torch.empty(B * Sq, H, dtype=BF16, device="cuda")
    dkv = torch.empty(B * Skv, 2 * H, dtype=BF16, device="cuda")
    delta = torch.empty(B, h, Sq, dtype=F32, device="cuda")
    k.attn_bwd(q, kk, v, add_mask, out, dout, lse, delta, dq, dkv[:, :H], dkv[:, H:], B, h, Sq, Skv, p_drop=p, seed=seed, tiled=tiled)
    assert rel_err(dq, qf.grad) < 3e-2
    assert rel_err(dkv[:, :H], kf.grad) < 3e-2
    assert rel_err(dkv[:, H:], vf.grad) < 3e-2
    dref = (oref.detach() * dout.float()).view(B, Sq, h, 64).sum(-1).permute(0, 2, 1)
    assert (delta - dref).abs().max().item() < 3e-2 * max(dref.abs().max().item(), 1.0)
    if p > 0:
        # keep bits: the forward leaves the dropout decisions of every (query, key) as bits -- the very decisions of the exported
        # mask -- and a backward that reads them (instead of hashing again) gives bitwise the gradients of the hashing one
        kb = k.attn_keepbits(B, h, Sq, Skv, "cuda")
        kb.fill_(-1)
        out2, lse2 = torch.empty_like(out), torch.empty_like(lse)
        k.attn_fwd(q, kk, v, add_mask, out2, lse2, B, h, Sq, Skv, p_drop=p, seed=seed, keepbits=kb, tiled=tiled)
        assert torch.equal(out2, out) and torch.equal(lse2, lse)
        wpl = (Skv + 127) // 128
        words = kb.view(B * h * Sq, 4, wpl).cpu().numpy().astype("uint32")
        import numpy as np
        key = np.arange(Skv)
        gi, wi, bi = (key % 16) // 4, ((key // 16) * 4 + key % 4) // 32, ((key // 16) * 4 + key % 4) % 32
        bits = (words[:, gi, wi] >> bi.astype("uint32")) & 1            # [rows, Skv]
        assert np.array_equal(bits.astype(bool), dmask.view(B * h * Sq, Skv).cpu().numpy() > 0)
        dq2, dkv2, delta2 = torch.empty_like(dq), torch.empty_like(dkv), torch.empty_like(delta)
        k.attn_bwd(q, kk, v, add_mask, out, dout, lse, delta2, dq2, dkv2[:, :H], dkv2[:, H:], B, h, Sq, Skv, p_drop=p, seed=seed,
                   keepbits=kb, tiled=tiled)
        assert torch.equal(dq2, dq) and torch.equal(dkv2, dkv) and torch.equal(delta2, delta)


def test_attention_fully_masked_row_matches_reference_softmax():
    """all keys masked (-10000 everywhere): the reference's softmax degenerates to softmax of the raw scores."""
    k = _k()
    B, h, S = 1, 1, 64
    q, kk, v = rnd(S, 64, seed=1), rnd(S, 64, seed=2), rnd(S, 64, seed=3)
    add_mask = torch.full((B, S), -10000.0, dtype=F32, device="cuda")
    out = torch.empty(S, 64, dtype=BF16, device="cuda"); lse = torch.empty(B, h, S, dtype=F32, device="cuda")
    k.attn_fwd(q, kk, v, add_mask, out, lse, B, h, S, S)
    s = q.float() @ kk.float().t() / 8.0 - 10000.0
    ref = torch.softmax(s, -1) @ v.float()
    assert rel_err(out, ref) < 1.5e-2


# ----------------------------------------------------------------------------------------------- embeddings
@pytest.mark.parametrize("B,S,H,p", [(2, 32, 128, 0.0), (8, 128, 768, 0.1)])
def test_embeddings(B, S, H, p):
    k = _k()
    V = 1000
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, V, (B, S), generator=g).cuda()
    ids[0, :5] = 0          # padding rows
    ids[1, :4] = 7          # repeated id
    tt = torch.randint(0, 2, (B, S), generator=g).cuda()
    word, pos, typ = rnd(V, H, seed=2, dtype=F32), rnd(512, H, seed=3, dtype=F32), rnd(2, H, seed=4, dtype=F32)
    gamma, beta = rnd(H, seed=5, dtype=F32) + 1.0, rnd(H, seed=6, dtype=F32)
    seed = 991
    mask = k.dropout_mask(B * S * H, p, seed, "cuda").view(B, S, H)
    y = torch.empty(B * S, H, dtype=BF16, device="cuda"); xhat = torch.empty_like(y)
    rstd = torch.empty(B * S, dtype=F32, device="cuda")
    k.embed_fwd(ids, tt, word, pos, typ, gamma, beta, y, xhat=xhat, rstd=rstd, p_drop=p, seed=seed)
    w, po, ty, ga, be = (t.clone().requires_grad_(True) for t in (word, pos, typ, gamma, beta))
    e = torch.nn.functional.embedding(ids, w, padding_idx=0) + po[:S][None] + ty[tt]
    mu = e.mean(-1, keepdim=True); var = ((e - mu) ** 2).mean(-1, keepdim=True)
    ref = (ga * ((e - mu) / torch.sqrt(var + 1e-12)) + be) * mask
    assert rel_err(y.view(B, S, H), ref) < 1e-2
    dy = rnd(B * S, H, seed=7)
    ref.backward(dy.float().view(B, S, H))
    dw, dp, dt = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    dg, db = torch.zeros_like(gamma), torch.zeros_like(beta)
    ws = k.ln_bwd_workspace(H, "cuda")
    k.embed_bwd(dy, ids, tt, xhat, rstd, gamma, dw, dp, dt, dg, db, ws, padding_idx=0, p_drop=p, seed=seed)
    assert dw[0].abs().max().item() == 0.0
    assert rel_err(dw, w.grad) < 2e-2
    assert rel_err(dp, po.grad) < 2e-2
    assert rel_err(dt, ty.grad) < 2e-2
    assert rel_err(dg, ga.grad) < 2e-2
    assert rel_err(db, be.grad) < 2e-2


# ----------------------------------------------------------------------------------------------- helpers
def test_elementwise_helpers():
    k = _k()
    x = torch.randn(1000003, device="cuda")
    xb = torch.empty(x.numel() + 5, dtype=BF16, device="cuda")[:x.numel()]
    k.cast_f32_to_bf16(x, xb)
    assert torch.equal(xb, x.to(BF16))
    back = torch.empty_like(x)
    k.cast_bf16_to_f32(xb, back)
    assert torch.equal(back, xb.float())
    # regions: both layouts
    B, R, Cc = 3, 49, 2048
    feats = torch.randn(B, Cc, 7, 7, device="cuda")
    tok = torch.empty(B * R, Cc, dtype=BF16, device="cuda")
    k.regions_to_tokens(feats, tok, B, R, Cc, 1)
    assert torch.equal(tok.view(B, R, Cc), feats.view(B, Cc, R).permute(0, 2, 1).to(BF16))
    f2 = torch.randn(B, 36, Cc, device="cuda")
    tok2 = torch.empty(B * 36, Cc, dtype=BF16, device="cuda")
    k.regions_to_tokens(f2, tok2, B, 36, Cc, 0)
    assert torch.equal(tok2.view(B, 36, Cc), f2.to(BF16))
    # colsum (bias gradients), incl. ragged N with a padded leading dimension
    for M, N, ld in [(4096, 768, 768), (4096, 2304, 2304), (300, 13, 16), (65, 3072, 3072)]:
        buf = rnd(M, ld, seed=M)
        xs = buf[:, :N]
        out = torch.ones(N, dtype=F32, device="cuda")
        k.colsum(xs, out, k.colsum_workspace(N, "cuda"), accumulate=True)
        assert rel_err(out, xs.float().sum(0) + 1.0) < 1e-4
    # dropout is its own backward; add; gate backward
    M, H = 513, 768
    a = rnd(M, H, seed=1); y = torch.empty_like(a); y2 = torch.empty_like(a)
    k.dropout(a, y, y2=y2, p_drop=0.1, seed=5)
    mask = k.dropout_mask(M * H, 0.1, 5, "cuda").view(M, H)
    assert rel_err(y, a.float() * mask) < 1e-2 and torch.equal(y, y2)
    b = rnd(M, H, seed=2); c = torch.empty_like(a)
    k.add_bf16(a, b, c)
    assert torch.equal(c, (a.float() + b.float()).to(BF16))
    dout, gte, cross, dci = rnd(M, H, seed=3), torch.sigmoid(rnd(M, H, seed=4).float()).to(BF16), rnd(M, H, seed=5), rnd(M, H, seed=6)
    du = torch.empty_like(a); dc = torch.empty_like(a)
    k.gate_bwd(dout, gte, cross, du, dc, dcross_in=dci)
    gf = gte.float()
    assert rel_err(du, dout.float() * cross.float() * gf * (1 - gf)) < 1e-2
    assert rel_err(dc, dout.float() * gf + dci.float()) < 1e-2


def test_token_ce():
    k = _k()
    M, Cn = 4096, 13
    logits = torch.randn(M, Cn, device="cuda") * 3
    labels = torch.randint(0, Cn, (M,), device="cuda")
    mask = (torch.rand(M, device="cuda") > 0.4).long()
    loss_sum = torch.zeros(1, device="cuda"); count = torch.zeros(1, device="cuda")
    dl = torch.empty(M, 16, dtype=BF16, device="cuda")
    k.token_ce(logits, labels, mask, loss_sum, count, dl)
    k.scale_by_ratio(dl, dl, den=count)
    lg = logits.clone().requires_grad_(True)
    tgt = torch.where(mask.bool(), labels, torch.full_like(labels, -100))
    ref = torch.nn.functional.cross_entropy(lg, tgt, ignore_index=-100)
    ref.backward()
    assert abs((loss_sum / count).item() - ref.item()) < 1e-4
    assert count.item() == mask.sum().item()
    assert rel_err(dl[:, :Cn], lg.grad) < 1e-2
    assert dl[:, Cn:].abs().max().item() == 0.0


@pytest.mark.parametrize("B,h,Sq,Skv,p", [(2, 12, 128, 36, 0.0), (2, 4, 128, 49, 0.1), (2, 2, 128, 128, 0.0),
                                          (1, 2, 49, 128, 0.0)])
def test_attention_fp8_forward(B, h, Sq, Skv, p):
    """BASELINE config c5: QK^T / PV on the fp8 (e4m3) matrix cores; diff against fp32 and against the bf16 kernel."""
    k = _k()
    H = h * 64
    q, kk, v = rnd(B * Sq, H, seed=1), rnd(B * Skv, H, seed=2), rnd(B * Skv, H, seed=3)
    add_mask = torch.zeros(B, Skv, dtype=F32, device="cuda")
    seed = 99
    dmask = k.attn_dropout_mask(B * h * Sq, Skv, p, seed, "cuda").view(B, h, Sq, Skv)
    o8 = torch.empty(B * Sq, H, dtype=BF16, device="cuda"); o16 = torch.empty_like(o8)
    l8 = torch.empty(B, h, Sq, dtype=F32, device="cuda"); l16 = torch.empty_like(l8)
    k.attn_fwd(q, kk, v, add_mask, o8, l8, B, h, Sq, Skv, p_drop=p, seed=seed, fp8=True)
    k.attn_fwd(q, kk, v, add_mask, o16, l16, B, h, Sq, Skv, p_drop=p, seed=seed)
    oref, lref = _attn_ref(q.float(), kk.float(), v.float(), add_mask, dmask, B, h, Sq, Skv)
    e8, e16 = rel_err(o8, oref), rel_err(o16, oref)
    print("\n[fp8 attention %dx%d p=%.1f] max rel err vs fp32: fp8 %.3e, bf16 %.3e; lse abs err fp8 %.3e"
          % (Sq, Skv, p, e8, e16, (l8 - lref).abs().max().item()))
    assert e8 < 0.12 and e16 < 1.5e-2        # e4m3 carries 3 mantissa bits: ~6 % per element before averaging
    assert (l8 - lref).abs().max().item() < 0.5
    with pytest.raises(RuntimeError):
        k.attn_fwd(rnd(200, 64), rnd(200, 64), rnd(200, 64), torch.zeros(1, 200, dtype=F32, device="cuda"),
                   torch.empty(200, 64, dtype=BF16, device="cuda"), None, 1, 1, 200, 200, fp8=True)


@pytest.mark.parametrize("M,N,Kd,act", [(32, 768, 768, 1), (1, 100, 256, 0), (50, 1024, 1024, 1), (64, 16, 128, 0)])
def test_linear_small_m(M, N, Kd, act):
    """A handful of rows through the operands-from-L2 kernel (BertPooler: tanh(dense(hidden[:, 0])))."""
    k = _k()
    S = 7
    full = rnd(M * S, Kd, seed=1, scale=0.5)
    x = full.view(M, S, Kd)[:, 0]                 # strided first-token rows, read in place
    W, bias = rnd(N, Kd, seed=2, scale=0.05), rnd(N, seed=3, dtype=F32)
    y = torch.empty(M, N, dtype=BF16, device="cuda")
    k.linear_small_m(x, W, bias, y, act=act)
    ref = x.float() @ W.float().t() + bias
    if act:
        ref = torch.tanh(ref)
    assert rel_err(y, ref) < 1e-2



def test_embed_bwd_rows_plus_scatter_equals_embed_bwd():
    """Row-sparse form of the embedding backward: per-token rows (zero for the padding id) + icka_embed_scatter_rows (f32 and
    bf16 rows, a scale, rows of several 'ranks' concatenated) == the dense scatter of icka_embed_bwd; every other output equal."""
    k = _k()
    B, S, H, V = 8, 32, 128, 200
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, V, (B, S), generator=g).cuda()
    ids[:, -5:] = 0
    tt = torch.randint(0, 2, (B, S), generator=g).cuda()
    dy = rnd(B * S, H, seed=5)
    xhat = rnd(B * S, H, seed=6)
    rstd = (torch.rand(B * S, generator=g) + 0.5).cuda()
    gamma = torch.randn(H, generator=g).cuda()
    lib = k._lib.load()
    ws = torch.empty(lib.icka_ln_bwd_workspace_floats(H), dtype=F32, device="cuda")
    outs = []
    for rows in (False, True):
        dword = torch.zeros(V, H, device="cuda")
        dpos, dtyp = torch.zeros(S, H, device="cuda"), torch.zeros(2, H, device="cuda")
        dg, db = torch.zeros(H, device="cuda"), torch.zeros(H, device="cuda")
        if rows:
            dtok = torch.full((B * S, H), float("nan"), device="cuda")
            k.embed_bwd_rows(dy, ids, tt, xhat, rstd, gamma, dtok, dpos, dtyp, dg, db, ws, vocab=V, p_drop=0.1, seed=9, accumulate=False)
            assert torch.isfinite(dtok).all() and dtok[(ids.view(-1) == 0)].abs().max().item() == 0.0
            k.embed_scatter_rows(dtok, ids.view(-1).contiguous(), dword, scale=1.0)
        else:
            k.embed_bwd(dy, ids, tt, xhat, rstd, gamma, dword, dpos, dtyp, dg, db, ws, p_drop=0.1, seed=9, accumulate=False)
        outs.append((dword, dpos, dtyp, dg, db))
    torch.cuda.synchronize()
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-5, atol=1e-6)
    assert outs[0][0][0].abs().max().item() == 0.0 and outs[1][0][0].abs().max().item() == 0.0
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert torch.equal(a, b)
    # two "ranks" concatenated, bf16 rows, scale 1/2
    two = torch.cat([dtok, 3.0 * dtok]).to(BF16)
    ids2 = torch.cat([ids.view(-1), ids.view(-1)]).contiguous()
    acc = torch.zeros(V, H, device="cuda")
    k.embed_scatter_rows(two, ids2, acc, scale=0.5)
    ref = torch.zeros(V, H, device="cuda").index_add_(0, ids2, two.float() * 0.5)
    ref[0] = 0.0
    assert torch.allclose(acc, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("M,N,K,p,res_dtype,twin", [(4096, 768, 768, 0.1, F32, "f32"), (4096, 768, 3072, 0.0, F32, "f32"),
                                                    (1024, 768, 768, 0.1, BF16, None), (4096, 1024, 1024, 0.1, F32, "f32"),
                                                    (2048, 1024, 4096, 0.0, torch.float16, "f16"),
                                                    # stripe counts that are not multiples of 8: a stripe's 8 blocks spread over
                                                    # several XCDs (M = 512 is the reference's test loop at batch 4)
                                                    (512, 768, 768, 0.1, F32, "f32"), (384, 768, 3072, 0.1, F32, "f32"),
                                                    (128, 1024, 1024, 0.0, F32, None), (3200, 768, 768, 0.1, F32, "f32")])
def test_gemm_ln_one_launch_equals_the_two_launches_bitwise(M, N, K, p, res_dtype, twin):
    """icka_gemm_ln (dense -> bias + dropout + residual -> LayerNorm in ONE launch: the 8 blocks of a 128-row stripe meet at an
    arrival counter, then finish 16 rows each through the row body shared with ln_fwd_kernel) == icka_gemm + icka_ln_fwd, BITWISE:
    y, twin, xhat, rstd and the f32 intermediate; repeated launches on ONE counter buffer (the counters reset themselves); the
    error word stays 0.  Shapes outside the fused kernel's domain return False and launch nothing.
    (reference: BertSelfOutput.forward Cross_Modal_Interaction_Module.py:561-565, BertOutput.forward :532-536)"""
    k = _k()
    h, w = rnd(M, K, seed=1, scale=0.5), rnd(N, K, seed=2, scale=0.5)
    bias, gamma, beta = rnd(N, seed=3, dtype=F32), rnd(N, seed=4, dtype=F32) + 1.0, rnd(N, seed=5, dtype=F32)
    res = rnd(M, N, seed=6, dtype=F32).to(res_dtype)
    seed = 0x1234567890

    def outs():
        o = torch.full((M, N), float("nan"), dtype=F32, device="cuda")
        y = torch.empty(M, N, dtype=BF16, device="cuda")
        tw = None if twin is None else torch.empty(M, N, dtype=F32 if twin == "f32" else torch.float16, device="cuda")
        return o, y, tw, torch.empty(M, N, dtype=BF16, device="cuda"), torch.empty(M, dtype=F32, device="cuda")

    tw_kw = lambda t: {} if twin is None else ({"y_f32": t} if twin == "f32" else {"y_f16": t})
    o0, y0, t0, xh0, rs0 = outs()
    k.gemm(k.GEMM_NT, h, w, o0)
    k.ln_fwd(o0, bias, res, gamma, beta, y0, xhat=xh0, rstd=rs0, eps=1e-12, p_drop=p, seed=seed, **tw_kw(t0))
    ref = torch.nn.functional.layer_norm((h.float() @ w.float().t() + bias) * k.dropout_mask(M * N, p, seed, "cuda").view(M, N)
                                         + res.float(), (N,), gamma, beta, 1e-12)
    assert rel_err(y0, ref) < 1e-2
    sync = k.gemm_ln_sync("cuda")
    for rep in range(3):        # one counter buffer, launch after launch
        o1, y1, t1, xh1, rs1 = outs()
        assert k.gemm_ln(h, w, o1, bias, res, gamma, beta, y1, sync, xhat=xh1, rstd=rs1, eps=1e-12, p_drop=p, seed=seed, **tw_kw(t1))
        torch.cuda.synchronize()
        assert torch.equal(o1, o0) and torch.equal(y1, y0) and torch.equal(xh1, xh0) and torch.equal(rs1, rs0), rep
        if twin is not None:
            assert torch.equal(t1, t0)
        k.gemm_ln_check_error("test")                                      # no stripe wait gave up
        assert int(sync.abs().sum().item()) == 0                           # counters back at zero
    # outside the domain: not 8 column tiles / rows not in whole 128-row stripes / more blocks than CUs -> False, nothing launched
    for (m2, n2) in ((4096, 2304), (200, 768), (8192, 768)):
        o = torch.zeros(m2, n2, dtype=F32, device="cuda")
        assert not k.gemm_ln(rnd(m2, K, seed=7), rnd(n2, K, seed=8), o, None, None, rnd(n2, seed=9, dtype=F32), rnd(n2, seed=10, dtype=F32),
                             torch.empty(m2, n2, dtype=BF16, device="cuda"), sync)
        assert o.abs().sum().item() == 0.0


@pytest.mark.parametrize("B,S,heads,K,p,f16,kb", [(32, 128, 12, 768, 0.1, False, False), (32, 128, 12, 768, 0.0, False, False),
                                                  (64, 128, 12, 768, 0.1, False, False), (32, 128, 16, 1024, 0.1, False, False),
                                                  (24, 128, 12, 768, 0.1, False, False), (32, 128, 12, 768, 0.1, True, False),
                                                  (32, 256, 16, 1024, 0.1, True, True), (32, 256, 16, 1024, 0.1, False, False),
                                                  (16, 256, 12, 768, 0.0, False, False)])
def test_gemm_qkv_attn_one_launch_equals_the_two_launches_bitwise(B, S, heads, K, p, f16, kb):
    """icka_gemm_qkv_attn (QKV projection + whole-head self-attention in ONE launch: every 256 x 192 tile of the 12-wave GEMM
    kernel = 256 / S samples x one head's q | k | v, the attention runs from the tile's LDS images) == icka_gemm +
    icka_attn_fwd_ex, BITWISE: the stacked qkv activation (the backward reads it), the context (and its fp16 copy), the
    log-sum-exp and the keep bits; with masked keys and with dropout (same counters: icka_attn_bwd regenerates the mask); bf16 and
    fp16 ("mixed16") operands; 128 and 256 tokens per sample.  (reference: BertSelfAttention.forward
    Cross_Modal_Interaction_Module.py:478-506)"""
    k = _k()
    H = 64 * heads
    M = B * S
    odt = torch.float16 if f16 else BF16
    x, w = rnd(M, K, seed=1, scale=0.5, dtype=odt), rnd(3 * H, K, seed=2, scale=0.08, dtype=odt)
    bias = rnd(3 * H, seed=3, dtype=F32)
    mask = torch.zeros(B, S, dtype=F32, device="cuda")
    for b in range(B):                       # ragged lengths: keys past the sample's length are masked
        mask[b, 20 + (37 * b) % (S - 20):] = -10000.0
    seed = 0x9876543210

    def outs():
        return (torch.full((M, 3 * H), float("nan"), dtype=BF16, device="cuda"), torch.full((M, H), float("nan"), dtype=BF16, device="cuda"),
                torch.full((B, heads, S), float("nan"), dtype=F32, device="cuda"),
                torch.full((M, H), float("nan"), dtype=torch.float16, device="cuda") if f16 else None,
                k.attn_keepbits(B, heads, S, S, "cuda").fill_(-1) if kb else None)

    q0, c0, l0, h0, b0 = outs()
    k.gemm(k.GEMM_NT, x, w, q0, bias=bias)
    k.attn_fwd(q0[:, :H], q0[:, H:2 * H], q0[:, 2 * H:], mask, c0, l0, B, heads, S, S, p_drop=p, seed=seed, out16=h0, keepbits=b0)
    for rep in range(2):
        q1, c1, l1, h1, b1 = outs()
        assert k.gemm_qkv_attn(x, w, bias, q1, mask, c1, l1, B, heads, S, p_drop=p, seed=seed, out16=h1, keepbits=b1)
        torch.cuda.synchronize()
        assert torch.equal(q1, q0), rep
        assert torch.equal(l1, l0), rep
        assert torch.equal(c1, c0), rep
        assert h0 is None or torch.equal(h1, h0)
        assert b0 is None or torch.equal(b1, b0)
    # eval form: no log-sum-exp wanted
    q1, c1, _, h1, _ = outs()
    assert k.gemm_qkv_attn(x, w, bias, q1, mask, c1, None, B, heads, S, p_drop=p, seed=seed, out16=h1)
    assert torch.equal(c1, c0) and torch.equal(q1, q0)
    # the backward takes what the fused forward left
    if p > 0:
        dctx = rnd(M, H, seed=11, scale=0.1)
        g0, g1 = torch.empty(M, 3 * H, dtype=BF16, device="cuda"), torch.empty(M, 3 * H, dtype=BF16, device="cuda")
        d0, d1 = torch.empty(B, heads, S, dtype=F32, device="cuda"), torch.empty(B, heads, S, dtype=F32, device="cuda")
        for (qq, cc, ll, gg, dd) in ((q0, c0, l0, g0, d0), (q1, c1, l0, g1, d1)):
            k.attn_bwd(qq[:, :H], qq[:, H:2 * H], qq[:, 2 * H:], mask, cc, dctx, ll, dd, gg[:, :H], gg[:, H:2 * H], gg[:, 2 * H:], B, heads,
                       S, S, p_drop=p, seed=seed)
        assert torch.equal(g0, g1)


def test_gemm_qkv_attn_declines_what_it_does_not_cover():
    """Outside the fused launch's domain (sequence length other than 128 / 256, rows that do not fill 256-row tiles, a grid of
    fewer than 128 tiles) the call returns False and writes nothing."""
    k = _k()
    for (B, S, heads) in ((64, 64, 12), (31, 128, 12), (4, 128, 12), (8, 384, 12)):
        H, M = 64 * heads, B * S
        x, w, bias = rnd(M, 768, seed=1), rnd(3 * H, 768, seed=2), rnd(3 * H, seed=3, dtype=F32)
        qkv, ctx = torch.zeros(M, 3 * H, dtype=BF16, device="cuda"), torch.zeros(M, H, dtype=BF16, device="cuda")
        assert not k.gemm_qkv_attn(x, w, bias, qkv, torch.zeros(B, S, dtype=F32, device="cuda"), ctx, None, B, heads, S)
        assert qkv.float().abs().sum().item() == 0.0 and ctx.float().abs().sum().item() == 0.0


def test_gemm_ln_stripe_wait_that_gives_up_is_never_silent(request):
    """The blocks of a 128-row stripe wait for each other inside icka_gemm_ln: if one never arrives (test hook: block 5 skips
    its arrival; small poll budget) the waits of that stripe give up -- bounded, no hang --, the rows finished from the
    incomplete stripe are NaN, the pinned error word makes the NEXT host touch-point raise, the counters are back at zero, and
    the next launch is healthy again.  (The two-launch path, icka_gemm + icka_ln_fwd, cannot fail this way; this is the price
    of the seam inside the launch.)"""
    k = _k()
    lib = k._lib.load()
    request.addfinalizer(lambda: lib.icka_gemm_ln_test_hooks(0, -1))
    M, N, K = 4096, 768, 768
    h, w = rnd(M, K, seed=1, scale=0.5), rnd(N, K, seed=2, scale=0.5)
    bias, gamma, beta = rnd(N, seed=3, dtype=F32), rnd(N, seed=4, dtype=F32) + 1.0, rnd(N, seed=5, dtype=F32)
    res = rnd(M, N, seed=6, dtype=F32)
    sync = k.gemm_ln_sync("cuda")
    o, y = torch.empty(M, N, dtype=F32, device="cuda"), torch.empty(M, N, dtype=BF16, device="cuda")
    assert k.gemm_ln(h, w, o, bias, res, gamma, beta, y, sync)
    torch.cuda.synchronize()
    good = y.clone()
    k.gemm_ln_check_error("healthy launch")
    lib.icka_gemm_ln_test_hooks(2000, 5)
    y.zero_()
    assert k.gemm_ln(h, w, o, bias, res, gamma, beta, y, sync)
    lib.icka_gemm_ln_test_hooks(0, -1)
    torch.cuda.synchronize()
    bad_rows = torch.isnan(y.float()).any(1)
    assert int(bad_rows.sum()) == 128                                       # exactly the stripe of the block that never arrived
    assert torch.equal(y[~bad_rows], good[~bad_rows])                       # every other stripe is complete and correct
    assert int(sync.abs().sum().item()) == 0                                # the counters reset themselves after the failure too
    with pytest.raises(k.GemmLnHandoffError):
        k.gemm_ln(h, w, o, bias, res, gamma, beta, y, sync)                 # next host touch-point raises (and clears the word)
    assert k.gemm_ln(h, w, o, bias, res, gamma, beta, y, sync)
    torch.cuda.synchronize()
    assert torch.equal(y, good)
    k.gemm_ln_check_error("after recovery")
