#!/bin/bash
# Same-box A/B of an environment knob on the bench step: alternates the given VAR=value settings, two rounds.
# usage: bash tools/ab_env.sh "VAR=a" "VAR=b" [-- bench args]
A=$1; B=$2; shift 2; [ "$1" = "--" ] && shift
ARGS="--steps 100 --warmup 10 --no-cpu-baseline --no-roofline --no-optimizer-leg --no-eager-leg $@"
for kv in "$A" "$B" "$A" "$B"; do
  env $kv timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s %8.3f ms/step %9.1f samples/s' % ('$kv', d['ms_per_step'], d['value']))" || exit 1
done
