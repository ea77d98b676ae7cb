#!/bin/bash
# The fused classifier head (passport_ops.pooled_linear: avg-pool + Linear in one launch per direction): parity, then the
# config-R step and the config-P shard with the library ops (DEEPIPR_OWN_HEAD=0) and with the kernel, alternating.
mkdir -p gpurun_out/r06ab
O=$GRAFT_REPO_ROOT/gpurun_out/r06ab
timeout 900 python -m pytest ${HEAD_AB_TESTS:-tests/test_head_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py tests/test_integration_gpu.py} -x -q -m gpu > $O/pytest.log 2>&1; tail -1 $O/pytest.log | cut -c1-200
for r in 1 2; do
  for v in 0 1; do
    export DEEPIPR_OWN_HEAD=$v
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_R_head${v}_$r.json
    python -c "import json; d=json.load(open('$O/bench_R_head${v}_$r.json')); print('own_head=$v round $r R', d['ms_per_step'], d['value'])"
    timeout 300 python bench.py --scheme 2 --classes 100 --batch 32 --steps 200 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_P_head${v}_$r.json
    python -c "import json; d=json.load(open('$O/bench_P_head${v}_$r.json')); print('own_head=$v round $r P', d['ms_per_step'], d['value'])"
  done
done
