# in-step phase stamps (prologue / loop / epilogue cycles, loader segments) of the N = 768 GEMM shapes of the c2 step
# needs the diagnostic library: make -C icka_amd/csrc EXTRA="-DICKA_GEMM_STAMP -DICKA_GEMM_ABLATE" OBJDIR=build_diag TARGET=../libicka_hip_diag.so
cd $GRAFT_REPO_ROOT
export ICKA_HIP_LIB=$GRAFT_REPO_ROOT/icka_amd/libicka_hip_diag.so
for f in "0,768,768" "0,768,3072" "1,768,3072" "1,768,2304" "1,768,768"; do
  ICKA_GEMM_STAMP_FILTER=$f timeout -k 10 200 python tools/bench_knob.py stamp=1 -- --steps 20 --warmup 5 --no-cpu-baseline --no-optimizer-leg --no-roofline > gpurun_out/r3_stamp.out 2> gpurun_out/r3_stamp.err || { tail -20 gpurun_out/r3_stamp.err; exit 1; }; grep "in-step stamps" gpurun_out/r3_stamp.err

done
