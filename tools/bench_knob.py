#!/usr/bin/env python3
"""A/B helper: run bench.py's main() in this process with launch-heuristic overrides -- given to the library the only way a
whole process can: the ICKA_TUNE_GEMM_* environment, read once when libicka_hip.so loads (there are no setters).
usage: python tools/bench_knob.py ring=4 tile_n=96 -- --steps 20 --warmup 5"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
cut = args.index("--") if "--" in args else len(args)
ENV = {"ring": "ICKA_TUNE_GEMM_RING", "tile_n": "ICKA_TUNE_GEMM_TILE_N", "ws": "ICKA_TUNE_GEMM_WARP_SPECIALIZED",
       "direct": "ICKA_TUNE_GEMM_DIRECT_EPILOGUE", "big": "ICKA_TUNE_GEMM_BIG_TILES", "w3grid": "ICKA_TUNE_GEMM_W3_GRID",
       "wide": "ICKA_TUNE_GEMM_WIDE_TILES", "ln_rows": "ICKA_TUNE_LN_ROWS_PER_WAVE"}
for kv in args[:cut]:
    k, v = kv.split("=")
    if k != "stamp":
        os.environ[ENV[k]] = str(int(v))
from icka_amd import _lib  # noqa: E402

lib = _lib.load()
stamp = None
for kv in args[:cut]:
    k, v = kv.split("=")
    if k == "stamp":   # build with EXTRA=-DICKA_GEMM_STAMP: every GEMM stamps into one buffer (last writer wins per block)
        import torch
        stamp = torch.zeros(8192, 16, dtype=torch.int64, device="cuda")
        lib.icka_diag_gemm_stamp_buffer.argtypes = [__import__("ctypes").c_void_p]
        lib.icka_diag_gemm_stamp_buffer(stamp.data_ptr())
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
