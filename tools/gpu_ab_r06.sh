#!/bin/bash
# Same-box A/B of a kernel change: A = csrc/libdeepipr_hip_head.so (the library built from the commit before; DEEPIPR_LIB), B = this
# tree.  Parity of the touched kernels first (AB_TESTS), then the config-R step and the ResNet50 batch-256 step, A and B alternating.
#   AB_TAG=r06v AB_TESTS="tests/test_conv_wino_gpu.py" AB_ROUNDS=2 bash tools/gpu_ab_r06.sh
TAG=${AB_TAG:-ab}
mkdir -p gpurun_out/$TAG
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
if [ -n "$AB_TESTS" ]; then
    timeout 900 python -m pytest $AB_TESTS -x -q -m gpu > $O/pytest.log 2>&1; tail -1 $O/pytest.log | cut -c1-200
fi
HEADLIB=$GRAFT_REPO_ROOT/deepipr_amd/csrc/libdeepipr_hip_head.so
for r in $(seq 1 ${AB_ROUNDS:-2}); do
  for v in head new; do
    unset DEEPIPR_LIB
    [ $v = head ] && export DEEPIPR_LIB=$HEADLIB
    if [ -z "$AB_NO_R" ]; then
      timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_R_${v}_$r.json
      python -c "import json; d=json.load(open('$O/bench_R_${v}_$r.json')); k=d.get('roofline_mfma_kernels',{}); print('$v $r R', d['ms_per_step'], d['value'], {n:r.get('avg_us') for n,r in k.items() if 'wino' in n} if isinstance(k,dict) else '')"
    fi
    if [ -z "$AB_NO_R50" ]; then
      timeout 400 python bench.py --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_r50_${v}_$r.json
      python -c "import json; d=json.load(open('$O/bench_r50_${v}_$r.json')); print('$v $r R50', d['ms_per_step'], d['value'])"
    fi
  done
done
