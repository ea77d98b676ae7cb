#!/bin/bash
# ConvBlock fusion threshold experiment (repeat runs: MIOpen's find mode makes single runs noisy)
mkdir -p gpurun_out; rm -f gpurun_out/fusemin.log
for rep in 1 2 3; do for fm in 1048576 0; do for cfg in "--arch alexnet --batch 64" "--scheme 2 --classes 100 --batch 32"; do
  echo "== rep$rep fuse_min=$fm $cfg" >> gpurun_out/fusemin.log
  DEEPIPR_CONVBLOCK_FUSE_MIN=$fm timeout 600 python bench.py $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/fusemin.log
done; done; done
cat gpurun_out/fusemin.log
