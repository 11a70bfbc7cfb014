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
