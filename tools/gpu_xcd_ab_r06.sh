#!/bin/bash
# A/B of the XCD-contiguous workgroup numbering (deepipr_conv_plan.h: dipr_xcd_contiguous) in one gpurun call:
#   A = csrc/libdeepipr_hip_head.so (the library built from the commit before the change; `DEEPIPR_LIB=`), B = this tree.
# Parity first (the kernels whose workgroup decode changed), then the per-shape 1x1 bench, then the two step times.
mkdir -p gpurun_out/r06q
O=$GRAFT_REPO_ROOT/gpurun_out/r06q
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_conv_1x1_gpu.py tests/test_conv_wino_gpu.py tests/test_conv_wgrad_gpu.py tests/test_conv_gemm_gpu.py -x -q -m gpu > $O/pytest_conv.log 2>&1
tail -2 $O/pytest_conv.log | cut -c1-300
HEADLIB=$GRAFT_REPO_ROOT/deepipr_amd/csrc/libdeepipr_hip_head.so
for v in head new new_nowino; do
    unset DEEPIPR_LIB DEEPIPR_XCD_REMAP
    [ $v = head ] && export DEEPIPR_LIB=$HEADLIB
    [ $v = new_nowino ] && export DEEPIPR_XCD_REMAP=0
    timeout 300 python tools/conv1x1_bench.py --no-check --json $O/conv1x1_$v.json > $O/conv1x1_$v.log 2>&1
    tail -1 $O/conv1x1_$v.log | cut -c1-300
    timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_R_$v.json
    python -c "import json,sys; d=json.load(open('$O/bench_R_$v.json')); print('$v R', d['ms_per_step'], d['value'])"
    timeout 400 python bench.py --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_r50_$v.json
    python -c "import json,sys; d=json.load(open('$O/bench_r50_$v.json')); print('$v R50', d['ms_per_step'], d['value'])"
done
