#!/bin/bash
# MIOpen solver experiment, finer than round 1's family toggles: switch off only the NHWC implicit-GEMM assembly
# solvers (the ones that wrap NCHW tensors in batched_transpose kernels), one direction at a time.
mkdir -p gpurun_out
OUT=gpurun_out/r02_miopen_solvers.log
: > $OUT
for cfg in "base=1" \
           "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0" \
           "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0" \
           "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0" \
           "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC=0 MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC=0"; do
  echo "== $cfg" >> $OUT
  env $cfg timeout 600 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-stress --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-230 >> $OUT
done
# kernel mix of the most promising variant
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof2 && env MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof2 -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-kernel-timing > /tmp/prof2.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof2/w_kernel_trace.csv --steps 20 --top 25 > $GRAFT_REPO_ROOT/gpurun_out/r02_miopen_wrw_off_steady_state.md
cd $GRAFT_REPO_ROOT
cat $OUT | cut -c1-200; head -30 gpurun_out/r02_miopen_wrw_off_steady_state.md | cut -c1-150
