#!/bin/bash
# the 1x1 GEMM's main loop without the exit from its middle: parity, per-shape times, the ResNet50 step
mkdir -p gpurun_out/r06r
O=$GRAFT_REPO_ROOT/gpurun_out/r06r
timeout 600 python -m pytest tests/test_conv_1x1_gpu.py tests/test_abi.py -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-300
timeout 300 python tools/conv1x1_bench.py --no-check --no-library --json $O/conv1x1.json > $O/conv1x1.log 2>&1; tail -1 $O/conv1x1.log | cut -c1-300
timeout 400 python bench.py --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_r50.json
python -c "import json; d=json.load(open('$O/bench_r50.json')); print('R50', d['ms_per_step'], d['value'])"
