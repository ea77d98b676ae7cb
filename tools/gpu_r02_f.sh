#!/bin/bash
# Round-2 GPU pass F: the whole parity suite and the smoke / bench contract on the round's final tree.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/r02_pytest_gpu_f.log
tail -4 gpurun_out/r02_pytest_gpu_f.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_f.json; cut -c1-260 gpurun_out/r02_bench_f.json
