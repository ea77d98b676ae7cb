#!/bin/bash
# ConvBlock fusion threshold under the DEFAULT launch mode (hipGraph replay): library norm + ReLU kernels below N
# elements vs the fused kernels everywhere.  Alternating repetitions (boxes and MIOpen's find mode make single runs
# noisy).  -> gpurun_out/${R}_fusemin.log
mkdir -p gpurun_out
R=${ROUND_TAG:-r03}
LOG=gpurun_out/${R}_fusemin.log
: > $LOG
one() { timeout 600 python bench.py "$@" --steps 100 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; }
for rep in 1 2; do
  for cfg in "--scheme 2 --classes 100 --batch 32" "--arch alexnet --batch 64" "--arch alexnet --scheme 2 --batch 64" "--batch 32"; do
    line="rep$rep [$cfg]"
    for fm in 1048576 262144 0; do
      line="$line  fuse_min=$fm: $(DEEPIPR_CONVBLOCK_FUSE_MIN=$fm one $cfg) ms"
    done
    echo "$line" | tee -a $LOG
  done
done
