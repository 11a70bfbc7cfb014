ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-eager-leg"
for kv in "X=0" "HIP_FORCE_DEV_KERNARG=1" "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "HSA_NO_SCRATCH_RECLAIM=1" "X=0" "GPU_MAX_HW_QUEUES=2"; do
  env $kv timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %8.3f ms/step (min %.3f max %.3f) %9.1f samples/s' % ('$kv', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['value']))" || echo "$kv failed"
done
