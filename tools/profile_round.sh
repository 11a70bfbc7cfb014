# (profiled runs skip the optimizer leg: its ATen kernels are not part of the step)
# Round profile set (run on the GPU box through gpurun): kernel stats of the default bench, the c4 / c5 presets, and the
# two PMC passes behind roofline.traffic.  Outputs under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -o c2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-optimizer-leg --no-eager-leg > $OUT/bench_c2_profiled.jsonl 2> $OUT/c2.err
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench_c2.jsonl 2>> $OUT/c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 30 --no-cpu-baseline --no-optimizer-leg --no-eager-leg > $OUT/bench_c4_profiled.jsonl 2> $OUT/c4.err
python3 $GRAFT_REPO_ROOT/bench.py --config c4 --steps 30 > $OUT/bench_c4.jsonl 2>> $OUT/c4.err
python3 $GRAFT_REPO_ROOT/bench.py --config c5 --steps 50 > $OUT/bench_c5.jsonl 2> $OUT/c5.err
# the other arithmetic mode of each: c2 in mixed16, c4 in pure bf16 (no CPU leg)
python3 $GRAFT_REPO_ROOT/bench.py --precision mixed16 --no-cpu-baseline > $OUT/bench_c2_mixed16.jsonl 2>> $OUT/c2.err
python3 $GRAFT_REPO_ROOT/bench.py --config c4 --precision bf16 --steps 30 --no-cpu-baseline > $OUT/bench_c4_bf16.jsonl 2>> $OUT/c4.err
# SURVEY section 8(f) rows at this commit, each with its algorithmic FLOP and fraction of the bf16 MFMA peak
python3 $GRAFT_REPO_ROOT/tools/tail_bench.py > $OUT/tail_gate1_bench.jsonl 2> $OUT/tail.err
python3 $GRAFT_REPO_ROOT/tools/tail_bench.py --with-encoder >> $OUT/tail_gate1_bench.jsonl 2>> $OUT/tail.err
python3 $GRAFT_REPO_ROOT/tools/cross_modal_bench.py > $OUT/cross_modal_bench.jsonl 2> $OUT/cross_modal.err
python3 $GRAFT_REPO_ROOT/tools/resnet_bench.py > $OUT/resnet_bench.jsonl 2> $OUT/resnet.err
python3 $GRAFT_REPO_ROOT/tools/lstm_bench.py > $OUT/lstm_bench.txt 2> $OUT/lstm.err
# the kernels of one replayed c2 step in order
rocprofv3 --kernel-trace --output-format csv -d $OUT/seq -o seq -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --repeats 1 --no-cpu-baseline --no-roofline --no-eager-leg --no-optimizer-leg > /dev/null 2> $OUT/seq.err
python3 $GRAFT_REPO_ROOT/tools/step_sequence.py $(find $OUT/seq -name "*kernel_trace.csv" | head -1) > $OUT/c2_step_sequence.txt 2>> $OUT/seq.err
# counter passes (MFMA busy / MOPS / FETCH_SIZE / WRITE_SIZE per kernel; c2 also yields gemm_traffic.json): a gpurun call of
# their own -- bash tools/profile_mfma.sh $TAG c2 && bash tools/profile_mfma.sh $TAG c4 -- the whole set exceeds one call's limit
cd $GRAFT_REPO_ROOT
# keep the merge-back small: stats + bench lines + traffic json only
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for f in $OUT/bench_*.jsonl; do echo $f; cut -c1-300 $f; done
