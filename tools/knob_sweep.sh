# step-level A/B of the GEMM library knobs on the c2 bench (same box, back to back)
mkdir -p gpurun_out; : > gpurun_out/knobs.log
for k in "ring=0" "ring=4" "ring=5" "tile_n=128" "ws=2" "direct=0" "big=1"; do
  echo "== $k" >> gpurun_out/knobs.log
  python tools/bench_knob.py $k -- --no-cpu-baseline --steps 60 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], d['value'], r['achieved'], r['avg_launch_us'])" >> gpurun_out/knobs.log
done
cat gpurun_out/knobs.log
