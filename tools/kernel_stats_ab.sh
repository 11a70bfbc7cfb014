# same-box per-kernel A/B (gpurun): the default bench under rocprofv3 --stats for each "<tree>[,ENV=VALUE]" given, e.g.
#   bash tools/kernel_stats_ab.sh _r02_tree . .,ICKA_ATTN_KEEPBITS=0
# prints the average duration of the heaviest kernels of the first run beside the others (names matched without the
# trailing template arguments that differ between trees).
I=0
for SPEC in "$@"; do
  T=${SPEC%%,*}; E=""; [ "$SPEC" != "$T" ] && E=${SPEC#*,}
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ab_$I; I=$((I+1))
  rm -rf $OUT; mkdir -p $OUT; echo "$SPEC" > $OUT/spec.txt
  cd /tmp && export TMPDIR=/tmp
  [ -n "$E" ] && export "$E"
  X=""; grep -q -- "--no-optimizer-leg" $GRAFT_REPO_ROOT/$T/bench.py && X="--no-optimizer-leg"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c2 -- python3 $GRAFT_REPO_ROOT/$T/bench.py --no-cpu-baseline --no-roofline $X --steps 60 > $OUT/bench.jsonl 2> $OUT/bench.err
  [ -n "$E" ] && unset "${E%%=*}"
  find $OUT -name "*kernel_trace.csv" -delete
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, os, re
def norm(k):
    k = k.replace("(anonymous namespace)::", "")
    k = re.sub(r"\(.*", "", k)
    k = re.sub(r"(attn_\w+<\d+, \d+, \w+)[^>]*>", r"\1>", k)
    k = re.sub(r"gemm_big_group_kernel<[^>]*>", "gemm_big_group_kernel", k)
    return k[:80]
runs = []
for d in sorted(glob.glob("gpurun_out/prof_ab_[0-9]*")):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
    if not f: continue
    tab = {}
    for r in csv.DictReader(open(f[0])):
        k = norm(r["Name"]); c, t = int(r["Calls"]), float(r["TotalDurationNs"])
        c0, t0 = tab.get(k, (0, 0.0)); tab[k] = (c0 + c, t0 + t)
    ms = ""
    try:
        import json; ms = json.loads(open(d + "/bench.jsonl").read().strip().splitlines()[-1])["ms_per_step"]
    except Exception: pass
    runs.append((open(d + "/spec.txt").read().strip(), tab, ms))
print("%-62s" % "kernel" + "".join("%26s" % r[0][-24:] for r in runs))
print("%-62s" % "ms_per_step (profiled)" + "".join("%26s" % r[2] for r in runs))
a = runs[0][1]
for k, v in sorted(a.items(), key=lambda kv: -kv[1][1])[:16]:
    line = "%-62s" % k[-62:]
    for _, tab, _ in runs:
        c, t = tab.get(k, (0, 0.0))
        line += "%8d x %7.2f us %+5.1f%%" % (c, t / c / 1e3 if c else 0, 100 * ((t / c) / (v[1] / v[0]) - 1) if c else 0)
    print(line)
PY
