#!/usr/bin/env python3
"""HBM/fabric-side traffic of the GEMM kernels of one c2 step: measured (profiles/r02_gemm_traffic.json, rocprofv3 PMC
passes) against (a) the algorithmic bytes (every operand once) and (b) the least an 8-XCD part can fetch.

Why (b): FETCH_SIZE counts what the eight private 4 MiB L2s pull from the fabric (Infinity Cache hits included).  Tiles
of one GEMM run on all 8 XCDs; with the tile grid cut pm x pn over the XCDs (pm * pn = 8), every XCD needs 1/pm of the
A panels and 1/pn of the B panels, so the chip fetches at least  min_{pm*pn=8} (pn*|A| + pm*|B|)  + epilogue operands,
whatever the tile order -- for the path's shapes (activations 6-25 MB, weights 1-5 MB) that is 2.5-4x the algorithmic
bytes.  Measured / (b) is the part a better tile order could still remove.

usage: python tools/traffic_model.py [profiles/r02_gemm_traffic.json]"""
import json
import sys

M, H, I = 4096, 768, 3072     # c2: tokens per step, hidden, intermediate


def bound(a_bytes, b_bytes):
    return min(pn * a_bytes + pm * b_bytes for pm, pn in ((8, 1), (4, 2), (2, 4), (1, 8)))


def mb(x):
    return x / 1e6


# (name, kernel key, launches per step, |A|, |B|, extra epilogue reads, write bytes)
bf, f4 = 2, 4
LAYERS, XL = 12, 1            # BERT layers + cross layer (the cross layer's q / kv projections are listed separately)
G = [
    ("qkv (NT 4096x2304x768)", "gemm_w3_kernel<false, false, 192>", LAYERS, M * H * bf, 3 * H * H * bf, 0, M * 3 * H * bf),
    ("ffn-up + GELU (NT 4096x3072x768)", "gemm_w3_kernel<false, false, 192>", LAYERS + XL, M * H * bf, I * H * bf, 0, 2 * M * I * bf),
    ("d(ffn-down) + GELU' (NN 4096x3072x768)", "gemm_w3_kernel<true, false, 192>", LAYERS + XL, M * H * bf, I * H * bf, M * I * bf,
     M * I * bf),
    ("out-proj (NT 4096x768x768, f32 out)", "gemm_ws_kernel<false, false, 3, 0, 96, false, false>", LAYERS + XL, M * H * bf, H * H * bf, 0,
     M * H * f4),
    ("ffn-down (NT 4096x768x3072, f32 out)", "gemm_ws_kernel<false, false, 3, 0, 96, false, false>", LAYERS + XL, M * I * bf, I * H * bf, 0,
     M * H * f4),
    ("d(ffn-up) + fan-in (NN 4096x768x3072)", "gemm_ws_kernel<false, true, 3, 0, 96, false, false>", LAYERS + XL, M * I * bf, I * H * bf,
     M * H * bf, M * H * bf),
    ("d(qkv) + fan-in (NN 4096x768x2304)", "gemm_ws_kernel<false, true, 3, 0, 96, false, false>", LAYERS, M * 3 * H * bf, 3 * H * H * bf,
     M * H * bf, M * H * bf),
    ("d(out-proj) (NN 4096x768x768)", "gemm_ws_kernel<false, true, 3, 0, 96, false, false>", LAYERS + XL, M * H * bf, H * H * bf, 0,
     M * H * bf),
]


def big_group_fetch():
    """Fetch of the grouped weight-gradient launch under its own tile order (csrc/gemm.hip gemm_big_group_kernel: XCD x takes
    the contiguous run x of the problems' concatenated lists of 256 x 128 tiles, each list ordered along its shorter side):
    every XCD fetches the distinct dY (256 x M x 2 B) and X (128 x M x 2 B) panels of its run.  Returns (tile panels, the
    second read of dY by the column-sum blocks of the same launch) in bytes."""
    probs = [(3 * H, H), (H, H), (I, H), (H, I)]            # dW[out, in] of qkv, out-proj, ffn-up, ffn-down
    tiles = []
    for pi, (mo, ni) in enumerate(probs):
        nbm, nbn = mo // 256, ni // 128
        for local in range(nbm * nbn):
            tm, tn = (local % nbm, local // nbm) if nbn > nbm else (local // nbn, local % nbn)
            tiles.append((pi, tm, tn))
    q, r = divmod(len(tiles), 8)
    pos, tot = 0, 0
    for x in range(8):
        n = q + (1 if x < r else 0)
        run = tiles[pos:pos + n]
        pos += n
        tot += len({(p, tm) for p, tm, _ in run}) * 256 * M * bf + len({(p, tn) for p, _, tn in run}) * 128 * M * bf
    return tot, sum(mo for mo, _ in probs) * M * bf


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02_gemm_traffic.json"
    meas = json.load(open(path))["per_kernel"]
    by_kernel = {}
    print("| GEMM | launches/step | algorithmic fetch MB | 8-XCD floor MB | write MB |")
    print("|---|---|---|---|---|")
    for name, key, n, a, b, extra, w in G:
        alg, fl = a + b + extra, bound(a, b) + extra
        print("| %s | %d | %.1f | %.1f | %.1f |" % (name, n, mb(alg), mb(fl), mb(w)))
        d = by_kernel.setdefault(key, [0, 0.0, 0.0, 0.0])
        d[0] += n; d[1] += n * alg; d[2] += n * fl; d[3] += n * w
    print()
    print("| kernel | launches/step (main shapes) | measured fetch MB/launch | algorithmic | 8-XCD floor | measured / floor | "
          "measured write MB | algorithmic write |")
    print("|---|---|---|---|---|---|---|---|")
    for key, (n, alg, fl, w) in by_kernel.items():
        mk = [k for k in meas if k.split("::")[-1].startswith(key) or key in k]
        m = meas[mk[0]]
        f_meas = m["fetch_MB_corrected"] / m["launches"]
        w_meas = m["write_MB"] / m["launches"]
        print("| `%s` | %d | %.1f | %.1f | %.1f | %.2f | %.1f | %.1f |"
              % (key, n, f_meas, mb(alg) / n, mb(fl) / n, f_meas / (mb(fl) / n), w_meas, mb(w) / n))
    g = [k for k in meas if "big_group" in k][0]
    ops = (2 * M * H + (M * H + M * I) * 2 + (3 * M * H + M * H)) * bf
    panels, colsum = big_group_fetch()
    f_meas = meas[g]["fetch_MB_corrected"] / meas[g]["launches"]
    print("| `gemm_big_group_kernel` (4 weight gradients of a layer + their column-sum blocks) | 13 | %.1f | %.1f | %.1f panels of "
          "its tile order + %.1f second read of dY by the column sums | %.2f | %.1f | %.1f |"
          % (f_meas, mb(ops), mb(panels), mb(colsum), f_meas / mb(panels + colsum), meas[g]["write_MB"] / meas[g]["launches"],
             mb((H * H + 2 * H * I + 3 * H * H) * f4)))


if __name__ == "__main__":
    main()
