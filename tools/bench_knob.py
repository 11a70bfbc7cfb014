#!/usr/bin/env python3
"""A/B helper: run bench.py's main() in this process after setting library knobs.
usage: python tools/bench_knob.py ring=4 tile_n=96 -- --steps 20 --warmup 5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from icka_amd import _lib  # noqa: E402

args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
lib = _lib.load()
stamp = None
for kv in args[:cut]:
    k, v = kv.split("=")
    if k == "stamp":   # build with EXTRA=-DICKA_GEMM_STAMP: every GEMM stamps into one buffer (last writer wins per block)
        import torch
        stamp = torch.zeros(8192, 16, dtype=torch.int64, device="cuda")
        lib.icka_gemm_set_stamp_buffer(stamp.data_ptr())
        continue
    rc = getattr(lib, {"ring": "icka_gemm_set_ring", "tile_n": "icka_gemm_set_tile_n", "ws": "icka_gemm_set_warp_specialized",
                       "direct": "icka_gemm_set_direct_epilogue", "big": "icka_gemm_set_big_tiles",
                       "w3grid": "icka_gemm_set_w3_grid"}[k])(int(v))
    assert rc == 0, (k, v, rc)
sys.argv = ["bench.py"] + args[cut + 1:]
import bench  # noqa: E402

bench.main()
if stamp is not None:
    b = stamp.double().cpu()
    b = b[b[:, 5] > 0]
    clk = b[:, 4] / b[:, 5] * 100.0
    per = b[:, :3].mean(0) / b[:, 6].mean()
    ph = b[:, 11:14].mean(0)
    print("in-step stamps (%s): %d blocks nk %.0f | clock median %.0f MHz (min %.0f max %.0f) | loader per k-tile: vmcnt-wait %.0f "
          "barrier %.0f issue %.0f | phases: prologue %.0f loop %.0f epilogue %.0f cycles"
          % (os.environ.get("ICKA_GEMM_STAMP_FILTER", "all"), b.shape[0], b[:, 6].mean().item(), clk.median().item(),
             clk.min().item(), clk.max().item(), per[0], per[1], per[2], ph[0], ph[1], ph[2]))
