cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention or gemm" > gpurun_out/r3_t10.log 2>&1 || { tail -30 gpurun_out/r3_t10.log; exit 1; }
tail -2 gpurun_out/r3_t10.log
for kb in 0 1; do for hh in 8 12 16; do timeout -k 10 120 python tools/attn_bench.py --iters 200 --shape 128,128,$hh --keepbits $kb || exit 1; done; done
bash tools/kernel_stats_ab.sh _r02_tree . .,ICKA_ATTN_KEEPBITS=0
