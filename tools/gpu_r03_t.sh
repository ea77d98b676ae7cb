#!/bin/bash
# shared first convolution behind the split of the V2 / V3 dual forward: V2 / V3 parity subset, then A/B
mkdir -p gpurun_out
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "private or v2 or V2 or v3 or shared_trunk or staged or dual or grouped" 2>&1 | grep -E "passed|failed|^FAILED|^E  " | cut -c1-200 | tail -8
COMMON="--steps 80 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing"
one() { python bench.py $COMMON "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2; do
  echo "V2 bs32 shared conv $(one --scheme 2 --classes 100 --batch 32)  off $(DEEPIPR_NO_SHARED_CONV=1 one --scheme 2 --classes 100 --batch 32)"
done
echo "alexnet V2 shared conv $(one --arch alexnet --scheme 2 --batch 64)  off $(DEEPIPR_NO_SHARED_CONV=1 one --arch alexnet --scheme 2 --batch 64)"
