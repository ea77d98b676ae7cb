#!/bin/bash
# MIOpen solver-family experiment: does the wgrad path without NHWC igemm (and its NCHW<->NHWC transposes) win?
mkdir -p gpurun_out
for cfg in "base" "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM=0" "MIOPEN_DEBUG_CONV_WINOGRAD=0" "MIOPEN_FIND_MODE=1"; do
  echo "== $cfg" >> gpurun_out/miopen.log
  if [ "$cfg" = base ]; then
    timeout 600 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-stress --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/miopen.log
  else
    env $cfg timeout 600 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-stress --no-kernel-timing 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/miopen.log
  fi
done
cat gpurun_out/miopen.log
