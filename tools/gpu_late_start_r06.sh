#!/bin/bash
# Late start of a CU's second workgroup in the Winograd forward / backward-data kernel (deepipr_conv_plan.h: dipr_late_start):
# parity of the kernels with it on, then the config-R step and the ResNet50 step per delay (DEEPIPR_WINO_STAGGER="ns,quarter rounds").
mkdir -p gpurun_out/r06t
O=$GRAFT_REPO_ROOT/gpurun_out/r06t
DEEPIPR_WINO_STAGGER=4000,4 timeout 600 python -m pytest tests/test_conv_wino_gpu.py -x -q -m gpu > $O/pytest_wino.log 2>&1; tail -1 $O/pytest_wino.log | cut -c1-200
for v in ${LATE_START_SWEEP:-0 1500,6 3000,6 5000,6 8000,6 3000,4 5000,4}; do
    export DEEPIPR_WINO_STAGGER=$v
    timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_R_$v.json
    python -c "import json; d=json.load(open('$O/bench_R_$v.json')); k=d.get('roofline_mfma_kernels',{}); print('$v R', d['ms_per_step'], d['value'], {n:(r.get('avg_us'), r.get('frac')) for n,r in k.items() if 'wino' in n} if isinstance(k,dict) else '')"
done
for v in ${LATE_START_R50:-0 3000,6 5000,6}; do
    export DEEPIPR_WINO_STAGGER=$v
    timeout 400 python bench.py --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_r50_$v.json
    python -c "import json; d=json.load(open('$O/bench_r50_$v.json')); print('$v R50', d['ms_per_step'], d['value'])"
done
