#!/usr/bin/env python3
"""The kernel sequence of ONE replayed c2 step from a rocprofv3 --kernel-trace CSV: everything between two bump_nonce kernels of the
timed region, with durations and the gap to the previous kernel.   usage: python tools/step_sequence.py <kernel_trace.csv> [which]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bump_nonce" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
lo, hi = idx[which], idx[which + 1]
prev_end = None
tot = gap_tot = 0.0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
    print("%7.2f us  gap %6.2f  grid %7s  %s" % ((e - s) / 1e3, gap, r["Grid_Size_X"], name))
    tot += (e - s) / 1e3
    gap_tot += max(gap, 0.0)
    prev_end = e
print("kernels %d, busy %.1f us, gaps %.1f us, span %.1f us" % (hi - lo, tot, gap_tot, (int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3))
