cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "gemm or linear" > gpurun_out/r3_t13.log 2>&1 || { tail -30 gpurun_out/r3_t13.log; exit 1; }
tail -2 gpurun_out/r3_t13.log
timeout -k 10 300 python tools/gemm_ab.py flagsync 0 1 4 || exit 1
for k in 0 4 1 0 4 1; do timeout -k 10 200 python tools/bench_knob.py flagsync=$k -- --steps 100 --warmup 10 --no-cpu-baseline --no-optimizer-leg > gpurun_out/r3_fs_$k.jsonl 2>/dev/null || exit 1; python3 -c "import json; d=json.loads(open(\"gpurun_out/r3_fs_$k.jsonl\").read().strip().splitlines()[-1]); print(\"flagsync $k\", d[\"ms_per_step\"], d[\"loss\"], d[\"roofline\"][\"achieved\"])"; done
