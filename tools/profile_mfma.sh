# MFMA-busy / MOPS / HBM-side counter passes over the eager bench step (run on the GPU box through gpurun).
# usage: bash tools/profile_mfma.sh <tag> <config>     -> gpurun_out/prof_<tag>/<config>_mfma_busy.json
TAG=${1:-r05}
CFG=${2:-c2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
STEPS=3; [ "$CFG" = "c4" ] && STEPS=2
B="$GRAFT_REPO_ROOT/bench.py --config $CFG --steps $STEPS --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-optimizer-leg --no-eager-leg"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${CFG}_busy -o b -- python3 $B > /dev/null 2> $OUT/${CFG}_busy.err &&
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/${CFG}_mops -o m -- python3 $B > /dev/null 2> $OUT/${CFG}_mops.err &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${CFG}_fetch -o f -- python3 $B > /dev/null 2> $OUT/${CFG}_fetch.err &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${CFG}_write -o w -- python3 $B > /dev/null 2> $OUT/${CFG}_write.err
cd $GRAFT_REPO_ROOT
cc() { find $OUT/$1 -name "*counter_collection.csv" | head -1; }
python3 tools/mfma_busy_from_pmc.py --busy $(cc ${CFG}_busy) --mops $(cc ${CFG}_mops) --fetch $(cc ${CFG}_fetch) --write $(cc ${CFG}_write) \
    --trace $(find $OUT/${CFG}_busy -name "*kernel_trace.csv" | head -1) --out $OUT/${CFG}_mfma_busy.json --label $CFG
if [ "$CFG" = "c2" ]; then
    python3 tools/traffic_from_pmc.py $(cc ${CFG}_fetch) $(cc ${CFG}_write) $OUT/gemm_traffic.json
fi
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
