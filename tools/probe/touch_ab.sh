cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or linear" > gpurun_out/r3_t19.log 2>&1 || { tail -30 gpurun_out/r3_t19.log; exit 1; }
tail -2 gpurun_out/r3_t19.log
for k in 0 5 0 5 3 8; do timeout -k 10 200 python tools/bench_knob.py touch=$k -- --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer-leg > gpurun_out/r3_touch_$k.jsonl 2>/dev/null || exit 1; python3 -c "import json; d=json.loads(open(\"gpurun_out/r3_touch_$k.jsonl\").read().strip().splitlines()[-1]); print(\"touch $k\", d[\"ms_per_step\"], d[\"loss\"], d[\"roofline\"][\"achieved\"], d[\"roofline\"][\"avg_launch_us\"])"; done
