#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, then WRITE_SIZE; they do not fit one pass on gfx950)
into the per-launch HBM traffic of the dominant kernel class (the MFMA GEMMs) -> profiles/<tag>_gemm_traffic.json.

Corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of
wide coalesced streaming reads (16 B/lane global_load and buffer_load...lds alike) -> doubled; WRITE_SIZE is exact for
16-B-per-lane stores.
usage: traffic_from_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import csv
import json
import sys


def per_kernel(path, counter):
    tot, n = {}, {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        n[k] = n.get(k, 0) + 1
    return tot, n


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    # (gemm_ln_kernel -- a dense GEMM with its LayerNorm phase in the same launch -- is listed per kernel below but kept out of the
    #  class: its bytes contain the LayerNorm's row traffic; bench.py reports it under roofline.fused_dense_ln)
    #  gemm_qkv_attn_kernel -- the QKV projection with its attention -- likewise: roofline.fused_qkv_attn)
    gemm = [k for k in fetch if "gemm" in k and "gemm_ln" not in k and "gemm_qkv_attn" not in k]
    fused = [k for k in fetch if "gemm_ln" in k]
    fused_qa = [k for k in fetch if "gemm_qkv_attn" in k]
    launches = sum(nf[k] for k in gemm)
    fb = sum(fetch[k] for k in gemm) * 1024.0 * 2.0
    wb = sum(write.get(k, 0.0) for k in gemm) * 1024.0
    out = {
        "kernel_class": "gemm_ws*/gemm_dma*/gemm_kernel (all MFMA GEMM launches of bench.py steps)",
        "launches": launches,
        "fetch_bytes_per_launch_corrected_x2": fb / launches,
        "write_bytes_per_launch": wb / max(sum(nw.get(k, 0) for k in gemm), 1),
        "traffic_bytes_per_launch": fb / launches + wb / max(sum(nw.get(k, 0) for k in gemm), 1),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes with --kernel-trace over "
                  "`python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline`; KiB -> bytes; "
                  "FETCH_SIZE doubled (gfx950 wide-read under-count)",
        "per_kernel": {k[:90]: {"launches": nf[k], "fetch_MB_corrected": fetch[k] * 2048.0 / 1e6,
                                "write_MB": write.get(k, 0.0) * 1024.0 / 1e6} for k in sorted(gemm)},
        "fused_dense_ln": {k[:90]: {"launches": nf[k], "fetch_MB_corrected": fetch[k] * 2048.0 / 1e6,
                                    "write_MB": write.get(k, 0.0) * 1024.0 / 1e6} for k in sorted(fused)},
        "fused_qkv_attn": {k[:90]: {"launches": nf[k], "fetch_MB_corrected": fetch[k] * 2048.0 / 1e6,
                                    "write_MB": write.get(k, 0.0) * 1024.0 / 1e6} for k in sorted(fused_qa)},
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("launches", "fetch_bytes_per_launch_corrected_x2", "write_bytes_per_launch",
                                          "traffic_bytes_per_launch")}))


if __name__ == "__main__":
    main()
