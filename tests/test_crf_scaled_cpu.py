"""CPU: the scaled linear-domain forward-backward that crf.hip runs for <= 16 tags (a_t = (a_{t-1} . E) x_t / sum(a_{t-1}),
normaliser lagging one step; bhat_{t-1} = E (x_t bhat_t / S_{t-1}); marginal_t = a_t . bhat_t), restated in numpy float32
and checked against the oracle (oracle/crf_oracle.py, float64, autograd) -- the algebra the kernel relies on, without a GPU."""
import numpy as np
import torch

from oracle import crf_oracle as O


def scaled_forward_backward(e, tags, L, start, end, trans):
    """one sample; returns (log Z, d(-llh)/d e [S, C]) in float32 arithmetic"""
    f = np.float32
    S, C = e.shape
    mT = trans.max()
    E = np.exp(trans - mT).astype(f)
    mx = e.max(1)
    x = np.exp(e - mx[:, None]).astype(f)
    mS = start.max()
    a = (np.exp(start - mS) * x[0]).astype(f)
    al, Sc = [a.copy()], {}
    lz = f(mS + mx[0])
    for t in range(1, S):
        if t < L:
            sm = a.sum(dtype=f)
            Sc[t] = sm
            a = ((a @ E) * (x[t] * (f(1) / sm))).astype(f)
            lz = f(lz + np.log(sm) + mx[t] + mT)
        al.append(a.copy())
    mE = end.max()
    zend = (a * np.exp(end - mE)).sum(dtype=f)
    logz = f(lz + mE + np.log(zend))
    bh = (np.exp(end - mE) / zend).astype(f)
    de = np.zeros((S, C), f)
    for t in range(S - 1, -1, -1):
        if t >= L and t > 0:
            continue
        marg = al[t] * bh
        onehot = np.zeros(C, f); onehot[tags[t]] = 1
        de[t] = marg - onehot          # d(-llh)/de
        if t == 0:
            break
        u = (x[t] * bh / Sc[t]).astype(f)
        bh = (E @ u).astype(f)
    return logz, de


def test_scaled_forward_backward_matches_the_oracle():
    g = torch.Generator().manual_seed(3)
    B, S, C = 5, 40, 13
    e = torch.randn(B, S, C, generator=g) * 3.0
    tags = torch.randint(0, C, (B, S), generator=g)
    lens = torch.tensor([S, 1, 17, 33, 2])
    mask = torch.arange(S)[None, :] < lens[:, None]
    start, end = torch.randn(C, generator=g) * 0.5, torch.randn(C, generator=g) * 0.5
    trans = torch.randn(C, C, generator=g) * 0.7
    trans[2, 5] = -1e4                                   # a forbidden transition: E = 0 exactly
    er = e.double().clone().requires_grad_(True)
    ref = O.crf_llh(er, tags, mask, start.double(), end.double(), trans.double())
    (-ref.sum()).backward()
    gold = O.crf_llh(e.double(), tags, mask, start.double(), end.double(), trans.double())
    for b in range(B):
        logz, de = scaled_forward_backward(e[b].numpy(), tags[b].numpy(), int(lens[b]), start.numpy(), end.numpy(), trans.numpy())
        # marginals (= gradient of -llh w.r.t. the emissions)
        assert np.abs(de - er.grad[b].numpy()).max() < 2e-4
        # normaliser: llh = gold-path score - log Z
        tb, L = tags[b].numpy(), int(lens[b])
        score = float(start[tb[0]] + e[b, 0, tb[0]] + sum(trans[tb[t - 1], tb[t]] + e[b, t, tb[t]] for t in range(1, L)) + end[tb[L - 1]])
        assert abs((score - float(logz)) - float(gold[b])) < 2e-3 * max(1.0, abs(float(gold[b])) / 50)
