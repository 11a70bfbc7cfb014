#!/bin/bash
# Per-site store-policy matrix: the c2 step with ordinary instead of streaming stores for the MAIN
# output of ONE producer site at a time (ICKA_GEMM_PLAIN_MASK, gemm.hip: site_bit), same box, baseline first / middle / last;
# then the union of the sites that beat the baseline.  usage (on the GPU box): bash tools/store_policy_matrix.sh [out.txt]
OUT=${1:-gpurun_out/store_policy_matrix.txt}
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-eager-leg"
run() {  # mask label
  ICKA_GEMM_PLAIN_MASK=$1 timeout -k 10 200 python bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('mask %-6s %-44s %8.3f ms/step %9.1f samples/s' % ('$1', '$2', d['ms_per_step'], d['value']))" | tee -a $OUT
}
: > $OUT
run 0x00 "baseline (all streaming)" || exit 1
run 0x01 "QKV -> attention" || exit 1
run 0x02 "out-proj / gate linears -> LayerNorm" || exit 1
run 0x04 "ffn-up (GELU) -> ffn-down" || exit 1
run 0x08 "ffn-down -> LayerNorm" || exit 1
run 0x00 "baseline (all streaming)" || exit 1
run 0x10 "d(ffn-down) dgrad -> d(ffn-up), wgrad" || exit 1
run 0x20 "d(ffn-up) dgrad -> LayerNorm bwd" || exit 1
run 0x40 "d(out-proj) dgrad -> attention bwd" || exit 1
run 0x80 "d(QKV) dgrad -> LayerNorm bwd" || exit 1
run 0x00 "baseline (all streaming)" || exit 1
run 0xff "all eight sites plain" || exit 1
