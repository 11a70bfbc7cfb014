#!/bin/bash
# A/B of two builds of the kernel library in ONE gpurun call (box-to-box variance is 1-2 %): alternates
# icka_amd/libicka_hip.so and the library given as $1, two rounds each.   usage: tools/ab_bench.sh icka_amd/libicka_hip_nt.so [bench args]
alt=$1; shift
for lib in icka_amd/libicka_hip.so "$alt" icka_amd/libicka_hip.so "$alt"; do
  ICKA_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d['roofline']['achieved'])" || exit 1
done
