"""GPU, two processes on the one MI355X of the test box, gloo: the data-parallel step with the REAL HIP backward
(ParamArena.flush_final call sites in ops.py / exact.py, bucket overlap bookkeeping) -- 2 ranks x half batch == 1 rank x
full batch (reference: apex DDP, My_cross_attention.py:768-776)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("precision,bar", [("bf16", 1e-2), ("fp32", 1e-5)])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, precision, bar):
    port = str(_free_port())
    outs = [str(tmp_path / ("r%d.pt" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), "2", port, outs[r], precision])
             for r in range(2)]
    try:
        for p in procs:
            assert p.wait(timeout=300) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r in range(2):
        res = torch.load(outs[r])
        print("\n[dp 2-rank %s rank %d] worst gradient rel-L2 vs single-rank full batch %.3e at %s; %d buckets, %d launched "
              "from inside backward" % (precision, r, res["worst"], res["key"], res["buckets"], res["overlapped"]))
        assert res["worst"] < bar, res
        assert res["buckets"] > 3 and res["overlapped"] >= res["buckets"] - 2, res


@pytest.mark.parametrize("comm,bar", [("f32", 1e-6), ("bf16", 6e-3)])
def test_segmented_step_matches_the_eager_step(tmp_path, comm, bar):
    """graph.SegmentedStep at world 1 over RCCL: linear graph segments + eager bucket all-reduces == plain eager step
    (f32 buckets: identical up to the all-reduce being an identity; bf16 buckets: one bf16 rounding of each gradient)."""
    out = str(tmp_path / "seg.pt")
    p = subprocess.Popen([sys.executable, os.path.join(HERE, "dp_segment_worker.py"), str(_free_port()), out, comm])
    try:
        assert p.wait(timeout=300) == 0
    finally:
        if p.poll() is None:
            p.kill()
    res = torch.load(out)
    print("\n[segmented step, %s buckets] %d segments for %d buckets (all-reduces after each: %s); worst gradient rel-L2 vs eager "
          "%.3e; loss %.5f (eager %.5f)" % (comm, res["segments"], res["buckets"], res["after"], res["worst"], res["loss"],
                                           res["ref_loss"]))
    assert res["segments"] > 3 and sum(res["after"]) == res["buckets"]
    assert res["worst"] < bar, res
    assert abs(res["loss"] - res["ref_loss"]) < 1e-6
