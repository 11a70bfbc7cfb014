# Quick per-kernel check of the c2 step on the GPU box (through gpurun): rocprofv3 kernel stats of one bench run, top kernels.
# usage (inside gpurun): bash tools/kernel_stats_check.sh [extra bench.py flags]
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_chk
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-optimizer-leg --no-cpu-baseline "$@" > $OUT/bench.jsonl 2> $OUT/bench.err
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_chk/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    print("%-84s %5s %7.2f us" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cut -c1-140 $OUT/bench.jsonl
