# world-1 overhead of the data-parallel machinery (bench.py --force-dist) for several bucket layouts
mkdir -p gpurun_out
: > gpurun_out/fd2.log
i=0
for cfg in "" "--force-dist --bucket-mb 32 --comm-f32" "--force-dist --bucket-mb 128 --comm-f32" "--force-dist --bucket-mb 128" "--force-dist --bucket-mb 512" "--force-dist --bucket-mb 128 --no-graph"; do
  i=$((i+1))
  echo "== $cfg" >> gpurun_out/fd2.log
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-roofline $cfg > gpurun_out/fd2_$i.out 2>>gpurun_out/fd2.err
  echo "rc=$?" >> gpurun_out/fd2.log
  grep '^{' gpurun_out/fd2_$i.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['launch'])" >> gpurun_out/fd2.log
done
cat gpurun_out/fd2.log
