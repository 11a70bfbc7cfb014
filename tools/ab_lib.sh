#!/bin/bash
# Same-box A/B of two builds of the kernel library on the bench step (alternating, two rounds each).
# usage: bash tools/ab_lib.sh <libA.so> <libB.so> [bench args]
A=$1; B=$2; shift 2
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-eager-leg $@"
for lib in "$A" "$B" "$A" "$B"; do
  ICKA_HIP_LIB=$lib timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-36s %8.3f ms/step %9.1f samples/s' % ('$lib', d['ms_per_step'], d['value']))" || exit 1
done
