#!/bin/bash
# grouped rank-2 update, second form (the group node returns the completed dW): parity + alternating A/B
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "grouped_rank2" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_round2_gpu.py -m gpu -q -x -p no:cacheprovider -k "graph or staged or golden or replay or private" 2>&1 | tail -6
COMMON="--steps 80 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing"
one() { python bench.py $COMMON "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "V1 grouped $(one)  per-layer $(DEEPIPR_NO_RANK2_BATCH=1 one)"
done
for rep in 1 2 3 4; do
  echo "V2 grouped $(one --scheme 2 --classes 100 --batch 32)  per-layer $(DEEPIPR_NO_RANK2_BATCH=1 one --scheme 2 --classes 100 --batch 32)"
done
echo "alexnet grouped $(one --arch alexnet --batch 64)  per-layer $(DEEPIPR_NO_RANK2_BATCH=1 one --arch alexnet --batch 64)"
