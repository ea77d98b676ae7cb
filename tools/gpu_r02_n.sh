#!/bin/bash
# Round-2 GPU pass N: bench.py --norm-type gn / in on the final tree.
mkdir -p gpurun_out
for nt in gn in; do
  timeout 600 python bench.py --norm-type $nt --no-cpu-baseline --no-stress 2>/dev/null | tail -1 > gpurun_out/r02_bench_cfg_resnet18_$nt.json
  cut -c1-200 gpurun_out/r02_bench_cfg_resnet18_$nt.json
done
