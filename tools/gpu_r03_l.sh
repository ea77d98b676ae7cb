#!/bin/bash
# Round 3: the grouped rank-2 update (one launch per layer group) -- parity, then A/B in the bench, then kernel list.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "grouped_rank2 or gamma_beta_multi" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "graph or staged or golden" 2>&1 | tail -8
COMMON="--steps 60 --warmup 15 --no-cpu-baseline --no-stress"
for rep in 1 2; do
  python bench.py $COMMON 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grouped   ', d['ms_per_step'], d['kernels']['gamma_beta_bwd'])"
  DEEPIPR_NO_RANK2_BATCH=1 python bench.py $COMMON 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('per layer ', d['ms_per_step'], d['kernels']['gamma_beta_bwd'])"
done
python bench.py $COMMON --scheme 2 --classes 100 --batch 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2 grouped   ', d['ms_per_step'], d['kernels']['gamma_beta_bwd'])"
DEEPIPR_NO_RANK2_BATCH=1 python bench.py $COMMON --scheme 2 --classes 100 --batch 32 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('V2 per layer ', d['ms_per_step'], d['kernels']['gamma_beta_bwd'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o l -- python $GRAFT_REPO_ROOT/bench.py --eager --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-kernel-timing > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/prof_l -name '*kernel_stats.csv' | head -1)
head -70 "$f" | cut -c1-150 > gpurun_out/r03_l_kernel_stats.csv
grep -i -E 'rank2|gamma_beta|copy|CUDAFunctor_add' gpurun_out/r03_l_kernel_stats.csv | cut -c1-140
