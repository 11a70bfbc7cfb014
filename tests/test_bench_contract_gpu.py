"""GPU: the contract of `python bench.py` at N = 1 (what the driver runs at the end of a round): stdout carries exactly ONE JSON
line with the metric / unit / config BASELINE.json names, whole-job throughput consistent with ms_per_step, the `roofline` object
(MFMA bound, achieved / peak / frac consistent, per-launch algorithmic bytes and flops, the committed PMC traffic and MFMA-busy
figures for c2) and the `cpu_baseline` object (the oracle timed on the host cores, bounded sample), plus the side legs this repo
reports beside the headline (eager launches, the wrapped module, host-resident inputs, the optimizer)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_keeps_the_contract():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--cpu-iters", "1",
                        "--optimizer-steps", "5", "--eager-steps", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    err = p.stderr.decode("utf-8", "replace")
    assert p.returncode == 0, err[-4000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"].startswith("MNER samples/sec (fwd+bwd) at seq=128, 36 regions, bs=32") and base["metric"].startswith(d["metric"][:55])
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16" and d["data"] == "synthetic"
    cfg = d["config"]
    assert cfg["workload"].startswith("c2:") and cfg["global_batch"] == 32 and cfg["seq_len"] == 128 and cfg["parallelism"] == "dp1"
    assert cfg["launch"] == "hipgraph" and "model" not in cfg
    assert d["value"] == pytest.approx(32 / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert 1.0 < d["ms_per_step"] < 50.0
    # the run's own noise bar: 5 blocks of `steps` steps, the headline from the median block
    assert d["repeats"] == 5 and len(d["ms_per_step_blocks"]) == 5
    assert d["ms_per_step_min"] <= d["ms_per_step"] <= d["ms_per_step_max"]
    assert sorted(d["ms_per_step_blocks"])[2] == pytest.approx(d["ms_per_step"], abs=2e-3)
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-4) and 0.02 < r["frac"] < 1.0
    assert r["achieved"] == pytest.approx(r["algorithmic_flop_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e12, rel=2e-2)
    # 125 GEMM launches per step: 26 of them (out-proj and ffn-down of the 12 + 1 layers) carry their LayerNorm in the same launch,
    # 12 (the QKV projections of the BERT layers) their self-attention
    f, fq = r["fused_dense_ln"], r["fused_qkv_attn"]
    assert r["launches_per_step"] + f["launches_per_step"] + fq["launches_per_step"] == 125
    assert f["launches_per_step"] == 26 and fq["launches_per_step"] == 12
    assert r["gemm_ms_per_step"] + f["ms_per_step"] + fq["ms_per_step"] < d["ms_per_step"]
    assert 10.0 < f["avg_launch_us"] < 80.0 and 10.0 < fq["avg_launch_us"] < 80.0
    assert r["traffic"] is None or r["traffic"] > r["algorithmic_bytes_per_launch"] * 0.9
    assert r["mfma_busy"] is None or 0.0 < r["mfma_busy"]["gemm_class_busy_frac_of_nominal_peak"] < 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "samples/s" and c["cores"] >= 1 and c["value"] > 0 and "batch 32" in c["sample"]
    assert d["gpu_over_cpu"] == pytest.approx(d["value"] / c["value"], rel=1e-2)
    # side legs: every one of them slower than (or, within noise, equal to) the replayed device-resident step
    assert d["eager_ms_per_step"] > 0.9 * d["ms_per_step"]
    assert d["wrapped_module_ms_per_step"] > 0.9 * d["ms_per_step"]
    assert d["host_inputs_ms_per_step"] > 0.9 * d["ms_per_step"] and d["host_inputs_prefetched_ms_per_step"] > 0.9 * d["ms_per_step"]
    assert d["host_input_bytes_per_step"] > 32 * 36 * 2048 * 4
    assert d["with_optimizer_ms_per_step"] > 0.95 * d["ms_per_step"] and 0 < d["refresh_us"] < 2000
    # (timing RELATIONS between the side legs are reported, not asserted: short legs on a shared box jitter)
    print("\n[bench contract] %.3f ms/step = %.0f samples/s; GEMM class %.3f of the bf16 roof; CPU oracle %.1f samples/s on %d threads"
          % (d["ms_per_step"], d["value"], r["frac"], c["value"], c["cores"]))
