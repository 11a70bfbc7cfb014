#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and HBM-side GB/s from rocprofv3 counter passes over bench.py -> profiles/<tag>_mfma_busy.json.

north_star asks for "rocprof HBM GB/s and MFMA-busy counters against the MI355X roofline".  Passes (each its own run, with
--kernel-trace only, program directly after `--`):
  busy  : --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  mops  : --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES     (optional)
  fetch : --pmc FETCH_SIZE            write : --pmc WRITE_SIZE                                    (optional)
  trace : the *_kernel_trace.csv of the busy pass (durations; a profiled pass -- never compared with un-profiled timings)

MfmaUtil per kernel = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs): the counter is per-SIMD busy
cycles summed over the chip's 1024 SIMDs (counter_defs.yaml: MfmaUtil = reduce(BUSY,sum) / (reduce(GUI_ACTIVE,max) x SIMD_NUM));
rocprofv3's GRBM_GUI_ACTIVE row is the SUM over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back) -> divided by 8.
An MFMA-bound kernel at roofline fraction f of the NOMINAL 2.4 GHz peak shows MfmaUtil ~= f x 2.4 GHz / effective clock.
HBM-side bytes: FETCH_SIZE x 2 (gfx950 wide-read under-count) + WRITE_SIZE, KiB -> bytes, as tools/traffic_from_pmc.py.

usage: mfma_busy_from_pmc.py --busy CSV [--mops CSV] [--fetch CSV] [--write CSV] --trace CSV --out JSON [--label c2]"""
import argparse
import csv
import json
import re

SIMDS = 1024
NOMINAL_GHZ = 2.4
XCDS = 8


def counters(path):
    """{kernel: {counter: [sum, n]}} over all dispatches of a counter_collection.csv"""
    out = {}
    for r in csv.DictReader(open(path)):
        k = out.setdefault(r["Kernel_Name"], {})
        c = k.setdefault(r["Counter_Name"], [0.0, 0])
        c[0] += float(r["Counter_Value"])
        c[1] += 1
    return out


def durations(path):
    out = {}
    for r in csv.DictReader(open(path)):
        d = out.setdefault(r["Kernel_Name"], [0.0, 0])
        d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        d[1] += 1
    return out


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:96]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--busy", required=True)
    ap.add_argument("--mops")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--trace", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--label", default="c2")
    a = ap.parse_args()
    busy = counters(a.busy)
    mops = counters(a.mops) if a.mops else {}
    fetch = counters(a.fetch) if a.fetch else {}
    write = counters(a.write) if a.write else {}
    dur = durations(a.trace)
    rows = {}
    tot_busy = tot_gui = 0.0
    cls = {}
    for k, c in busy.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        b, n = c["SQ_VALU_MFMA_BUSY_CYCLES"]
        gui = c["GRBM_GUI_ACTIVE"][0] / XCDS
        if gui <= 0:
            continue
        ns, nd = dur.get(k, [0.0, 0])
        row = {"launches": n, "mfma_util_pct": round(100.0 * b / (gui * SIMDS), 2),
               # the same busy cycles against the cycles the NOMINAL 2.4 GHz clock offers in the kernel's traced duration:
               # directly comparable with FLOP / time / 2.5 PFLOP/s (1024 SIMDs x 1024 bf16 FLOP per SIMD-cycle x 2.4 GHz);
               # GUI_ACTIVE spans more than the dispatch on launches well under 0.3 ms, so mfma_util_pct reads LOW there
               "mfma_busy_frac_of_nominal_peak": round(b / (SIMDS * ns * NOMINAL_GHZ), 4) if ns > 0 else None,
               "avg_us_profiled": round(ns / max(nd, 1) / 1e3, 2), "avg_gui_cycles_per_xcd": round(gui / n),
               "gui_cycles_over_duration_ghz": round(gui / ns, 3) if ns > 0 else None}
        m = mops.get(k, {})
        if m:
            ops = 512.0 * (m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0, 0])[0] + m.get("SQ_INSTS_VALU_MFMA_MOPS_F16", [0, 0])[0])
            nm = max(m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", [0, 1])[1], 1)
            row["mfma_gflop_per_launch_counted"] = round(ops / nm / 1e9, 3)
            if nd:
                row["counted_tflops"] = round(ops / nm / (ns / nd) / 1e3, 1)          # FLOP / ns / 1e3 = TFLOP/s
                row["counted_frac_of_2500"] = round(ops / nm / (ns / nd) / 1e3 / 2500.0, 4)
            if "SQ_BUSY_CU_CYCLES" in m:
                row["sq_busy_cu_quadcycles_per_launch"] = round(m["SQ_BUSY_CU_CYCLES"][0] / m["SQ_BUSY_CU_CYCLES"][1])
        if k in fetch or k in write:
            fb = fetch.get(k, {}).get("FETCH_SIZE", [0.0, 1])
            wb = write.get(k, {}).get("WRITE_SIZE", [0.0, 1])
            per = fb[0] / max(fb[1], 1) * 2048.0 + wb[0] / max(wb[1], 1) * 1024.0
            row["hbm_side_MB_per_launch"] = round(per / 1e6, 3)
            if nd:
                row["hbm_side_GBps"] = round(per / (ns / nd), 1)      # bytes / ns = GB/s
                row["hbm_frac_of_8TBps"] = round(per / (ns / nd) / 8000.0, 3)
        rows[short(k)] = row
        kind = ("gemm+layernorm (one launch)" if "gemm_ln" in k else "qkv gemm+attention (one launch)" if "gemm_qkv_attn" in k else "gemm" if "gemm" in k else "attention" if "attn" in k else "layernorm" if ("ln_" in k or "embed" in k) else
                "lstm" if "lstm" in k else "other")
        cc = cls.setdefault(kind, [0.0, 0.0, 0.0])
        cc[0] += b
        cc[1] += gui
        cc[2] += ns
        tot_busy += b
        tot_gui += gui
    out = {
        "label": a.label,
        "method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 3 --warmup 1 "
                  "--no-graph --no-cpu-baseline --no-roofline --no-optimizer-leg --no-eager-leg (+ separate passes for the "
                  "MOPS / FETCH_SIZE / WRITE_SIZE columns); MfmaUtil = BUSY / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                  "mfma_busy_frac_of_nominal_peak = BUSY / (1024 SIMDs x traced duration x 2.4 GHz); durations are the profiled "
                  "pass's own",
        "classes": {k: {"mfma_util_pct": round(100.0 * v[0] / (v[1] * SIMDS), 2),
                        "mfma_busy_frac_of_nominal_peak": round(v[0] / (SIMDS * v[2] * NOMINAL_GHZ), 4) if v[2] > 0 else None,
                        "share_of_kernel_time_pct": None} for k, v in cls.items()},
        "all_kernels_mfma_util_pct": round(100.0 * tot_busy / (tot_gui * SIMDS), 2) if tot_gui else None,
        "per_kernel": dict(sorted(rows.items(), key=lambda kv: -kv[1]["avg_us_profiled"] * kv[1]["launches"])),
    }
    tot_ns = sum(v[2] for v in cls.values())
    for k, v in cls.items():
        out["classes"][k]["share_of_kernel_time_pct"] = round(100.0 * v[2] / tot_ns, 1) if tot_ns else None
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({"classes": out["classes"], "all": out["all_kernels_mfma_util_pct"]}))


if __name__ == "__main__":
    main()
