#!/usr/bin/env python3
"""Diagnostic: per-kernel register / scratch / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py icka_amd/csrc/gemm.hip [name-filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + [a for a in sys.argv[3:]]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    print("%-100s vgpr %3d agpr %3d spill %3d scratch %4d occ %d lds %6d" % (
        name[:100], v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("VGPRs Spill", -1),
        v.get("ScratchSize [bytes/lane]", -1), v.get("Occupancy [waves/SIMD]", -1), v.get("LDS Size [bytes/block]", -1)))
