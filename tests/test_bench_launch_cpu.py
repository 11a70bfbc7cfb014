"""CPU: `python bench.py --gpus 2` without a launcher becomes the launcher (bench.self_launch): it starts the ranks as
children of a `python -m torch.distributed.run` child and relays their exit status.  Without a GPU the ranks stop at
bench.py's own "needs an MI355X" check -- which is exactly what this test looks for: the parent never raises the round-2
"launch with: python -m torch.distributed.run ..." refusal, and the ranks' failure comes back as a non-zero status."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_self_launches_and_relays_the_ranks_status():
    if torch.cuda.is_available():      # on a GPU box the GPU-marked test runs the same flow for real
        return
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=300)
    err = p.stderr.decode("utf-8", "replace")
    assert "self-launch" in err and "torch.distributed.run" in err, err[-2000:]
    assert "needs an MI355X" in err, err[-2000:]         # both ranks started and ran bench.main()
    assert "launch with:" not in err
    assert p.returncode != 0
    assert p.stdout.decode().strip() == ""


def test_self_launch_stops_ranks_that_never_finish(tmp_path):
    """bench.self_launch gives the ranks a process group of their own and a deadline (ICKA_BENCH_LAUNCH_TIMEOUT): ranks that hang
    are stopped as a group and the parent returns 124 instead of waiting for ever."""
    hang = tmp_path / "hang.py"
    hang.write_text("import time, sys\nprint('rank up', file=sys.stderr, flush=True)\ntime.sleep(600)\n")
    code = (
        "import os, sys\n"
        "sys.path.insert(0, %r)\n"
        "import bench\n"
        "bench.__file__ = %r\n"           # the module the launcher starts as a rank
        "sys.argv = ['bench.py']\n"
        "sys.exit(bench.self_launch(2, 1))\n" % (ROOT, str(hang)))
    env = dict(os.environ, ICKA_BENCH_LAUNCH_TIMEOUT="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # (the ranks would sleep for 600 s: the run's own timeout below is the proof that they were stopped)
    p = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=400)
    err = p.stderr.decode("utf-8", "replace")
    assert p.returncode == 124, (p.returncode, err[-2000:])
    assert "stopping the ranks" in err, err[-2000:]


def test_bench_gpus_8_launch_flow_dry_run():
    """`python bench.py --gpus 8` at the world size BASELINE's c3 names, as far as a box without 8 GPUs can
    prove it (ICKA_BENCH_DRY=1: gloo, the step is a sleep): eight ranks start under the self-launcher, rank r maps to device r
    (My_cross_attention.py:653-657), the timing is the max over ranks, rank 0 alone runs its extra legs while the others wait,
    and exactly one JSON line comes back with the whole-job value."""
    import json
    env = dict(os.environ, ICKA_BENCH_DRY="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    err = p.stderr.decode("utf-8", "replace")
    assert p.returncode == 0, err[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, (lines, err[-2000:])
    out = json.loads(lines[0])
    assert out["dry"] and out["n_gpus"] == 8 and out["config"]["global_batch"] == 256 and out["config"]["parallelism"] == "dp8"
    assert out["rank_devices"] == [[r, r] for r in range(8)]
    assert out["ms_per_step"] >= 2.0 + 0.5 * 7 - 0.2          # the slowest rank's sleep: max over ranks, not rank 0's
    assert out["value"] == __import__("pytest").approx(256 / (out["ms_per_step"] * 1e-3), rel=1e-3)
