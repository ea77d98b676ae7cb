#!/bin/bash
# Round 3: full -m gpu suite with the atomic solver switched off for the session; ConvBlock fusion threshold under graph
# replay; two gloo ranks sharing the GPU (launch-form fallback down to "in-launch exchange off").
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_gpu_7.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu_7.log
grep -E "passed|failed|^FAILED" gpurun_out/r03_pytest_gpu_7.log | tail -5
ROUND_TAG=r03 bash tools/gpu_fusemin.sh
: > gpurun_out/r03_two_ranks_one_gpu.jsonl
DEEPIPR_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-stress --no-kernel-timing \
    2> gpurun_out/r03_two_ranks.err | grep '"metric"' >> gpurun_out/r03_two_ranks_one_gpu.jsonl
echo "two ranks rc=$?"; grep "given up" gpurun_out/r03_two_ranks.err | cut -c1-200
python - <<'PY'
import json
for l in open('gpurun_out/r03_two_ranks_one_gpu.jsonl'):
    d = json.loads(l)
    print(d['n_gpus'], d.get('world_size_seen'), d['value'], d['ms_per_step'], d.get('exchange_us_exposed'), d.get('exchange_timeouts'), d['config']['launch'])
PY
