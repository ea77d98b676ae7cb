#!/bin/bash
# Two K ranges per workgroup in the Winograd weight gradient (k_conv_wino_wgrad2; DEEPIPR_WGRAD_PAIR=0 restores one): parity of
# every weight-gradient test, then the config-R step and the config-P shard with and without, alternating.
mkdir -p gpurun_out/r06ac
O=$GRAFT_REPO_ROOT/gpurun_out/r06ac
timeout 1200 python -m pytest ${PAIR_AB_TESTS:-tests/test_conv_wgrad_gpu.py tests/test_models_gpu.py tests/test_parity_gpu.py} -x -q -m gpu > $O/pytest.log 2>&1; tail -1 $O/pytest.log | cut -c1-200
for r in 1 2; do
  for v in 0 1; do
    export DEEPIPR_WGRAD_PAIR=$v
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_R_pair${v}_$r.json
    python -c "import json; d=json.load(open('$O/bench_R_pair${v}_$r.json')); k=d.get('roofline_mfma_kernels',{}); print('pair=$v round $r R', d['ms_per_step'], d['value'], k.get('conv_wino_wgrad',{}).get('avg_us'), d['roofline'].get('reduce_us_per_step_all_wgrads'))"
    timeout 300 python bench.py --scheme 2 --classes 100 --batch 32 --steps 200 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_P_pair${v}_$r.json
    python -c "import json; d=json.load(open('$O/bench_P_pair${v}_$r.json')); print('pair=$v round $r P', d['ms_per_step'], d['value'])"
  done
done
