"""CPU: statistics of the counter-hash dropout RNG (icka_amd/csrc/common.h: icka_hash / drop_pair), restated in numpy.
The attention sites draw TWO keep decisions from one 32-bit hash (low / high 16 bits against p * 2^16): the two halves must be
as good as two separate hashes -- keep rate, no correlation between the halves, between neighbouring pairs, or between two
dropout sites (seeds) at the same element index."""
import numpy as np

C0 = np.uint32(0x9E3779B1)


def icka_hash(s0, s1, idx):
    with np.errstate(over="ignore"):
        x = (idx.astype(np.uint32) * C0 + np.uint32(s0)).astype(np.uint32)
        x ^= x >> np.uint32(16); x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
        x ^= x >> np.uint32(15); x = (x * np.uint32(0x846ca68b) + np.uint32(s1)).astype(np.uint32)
        x ^= x >> np.uint32(16)
    return x


def keep_pair(seed, pidx, p):
    h = icka_hash(seed & 0xffffffff, seed >> 32, pidx)
    t = np.uint32(int(p * 4294967296.0)) >> np.uint32(16)
    return (h & np.uint32(0xffff)) >= t, (h >> np.uint32(16)) >= t


def corr(a, b):
    a = a.astype(np.float64) - a.mean(); b = b.astype(np.float64) - b.mean()
    return float((a * b).mean() / np.sqrt((a * a).mean() * (b * b).mean()))


def test_pair_decisions_keep_rate_and_independence():
    n, p = 1 << 21, 0.1
    idx = np.arange(n, dtype=np.uint32)
    for seed in (0x1234_5678_9abc, 0x0bad_cafe_1234_5678, 1):
        even, odd = keep_pair(seed, idx, p)
        assert abs(even.mean() - (1 - p)) < 1.5e-3 and abs(odd.mean() - (1 - p)) < 1.5e-3
        bar = 4.0 / np.sqrt(n)   # ~4 sigma of the sample correlation of independent bits
        assert abs(corr(even, odd)) < bar                     # the two halves of one hash
        assert abs(corr(even[:-1], even[1:])) < bar           # neighbouring pairs
        assert abs(corr(odd[:-1], even[1:])) < bar
        assert abs(corr(even[:-64], even[64:])) < bar         # one attention row further (Skv = 128)
    e1, o1 = keep_pair(0x1111_2222_3333, idx, p)
    e2, o2 = keep_pair(0x1111_2222_3334, idx, p)              # two sites: consecutive seeds, same element indices
    assert abs(corr(e1, e2)) < 4.0 / np.sqrt(n) and abs(corr(o1, o2)) < 4.0 / np.sqrt(n)


def test_flat_decisions_keep_rate():
    n, p = 1 << 20, 0.1
    idx = np.arange(n, dtype=np.uint32)
    h = icka_hash(0x9abc, 0x1234_5678, idx)
    keep = h >= np.uint32(int(p * 4294967296.0))
    assert abs(keep.mean() - (1 - p)) < 1.5e-3
    assert abs(corr(keep[:-1], keep[1:])) < 4.0 / np.sqrt(n)
