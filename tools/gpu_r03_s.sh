#!/bin/bash
# Round 3, closing call: full -m gpu suite on the final tree, the headline bench line and the config-P shard line.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_gpu_8.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu_8.log
grep -E "passed|failed|^FAILED" gpurun_out/r03_pytest_gpu_8.log | tail -5
timeout 600 python bench.py > gpurun_out/r03_bench_n1_final.json 2> gpurun_out/r03_bench_n1_final.err
python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_n1_final.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
timeout 600 python bench.py --scheme 2 --classes 100 --batch 32 --steps 60 --warmup 15 > gpurun_out/r03_bench_cfg_P_shard.json 2>/dev/null
python -c "
import json; d=json.loads(open('gpurun_out/r03_bench_cfg_P_shard.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
