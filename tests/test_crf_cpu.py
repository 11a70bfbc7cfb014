"""CPU: the CRF oracle (restated pytorch-crf 0.7.2 algorithm) against brute-force enumeration of every tag path.
The third-party package itself is absent (parity with it is unpinned); this pins the restatement's own arithmetic."""
import torch

from oracle import crf_oracle as C


def _params(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(c, generator=g) - 0.5, torch.rand(c, generator=g) - 0.5, torch.rand(c, c, generator=g) - 0.5)


def test_llh_and_decode_match_enumeration():
    Cn, S, B = 3, 5, 4
    start, end, trans = _params(Cn, 1)
    g = torch.Generator().manual_seed(2)
    e = torch.randn(B, S, Cn, generator=g)
    tags = torch.randint(0, Cn, (B, S), generator=g)
    lens = torch.tensor([5, 3, 1, 4])
    mask = (torch.arange(S)[None, :] < lens[:, None])
    llh = C.crf_llh(e, tags, mask, start, end, trans)
    paths = C.crf_decode(e, mask, start, end, trans)
    for b in range(B):
        L = int(lens[b])
        logz, best, _ = C.brute_force(e[b], L, start, end, trans)
        gold = start[tags[b, 0]] + e[b, 0, tags[b, 0]]
        for i in range(1, L):
            gold = gold + trans[tags[b, i - 1], tags[b, i]] + e[b, i, tags[b, i]]
        gold = gold + end[tags[b, L - 1]]
        assert abs(float(llh[b]) - float(gold - logz)) < 1e-5
        assert paths[b] == best
    assert abs(float(C.crf_reduce(llh, mask, "token_mean")) - float(llh.sum() / lens.sum())) < 1e-6
    assert abs(float(C.crf_reduce(llh, mask, "mean")) - float(llh.mean())) < 1e-6


def test_no_mask_means_all_on():
    start, end, trans = _params(4, 3)
    e = torch.randn(2, 6, 4, generator=torch.Generator().manual_seed(4))
    tags = torch.randint(0, 4, (2, 6), generator=torch.Generator().manual_seed(5))
    a = C.crf_llh(e, tags, None, start, end, trans)
    b = C.crf_llh(e, tags, torch.ones(2, 6, dtype=torch.bool), start, end, trans)
    assert torch.equal(a, b)
