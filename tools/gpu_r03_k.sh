#!/bin/bash
# Round 3, final evidence on the final tree: full -m gpu suite FIRST (fresh box), then the rehearsal table, then bench.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_gpu_4.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu_4.log
tail -5 gpurun_out/r03_pytest_gpu_4.log
ROUND_TAG=r03 timeout 900 bash tools/gpu_ddp1.sh
timeout 600 python bench.py > gpurun_out/r03_bench_n1_final.json 2> gpurun_out/r03_bench_n1_final.err
tail -c 1500 gpurun_out/r03_bench_n1_final.json
