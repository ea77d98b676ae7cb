#!/bin/bash
# GPU round script: parity tests, smoke, bench, kernel trace.  Run via gpurun from the repo root.
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/gpu.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/gpu.txt 2>/dev/null
python -c "import os;print('affinity',len(os.sched_getaffinity(0)))" >> gpurun_out/gpu.txt
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /tmp/prof.log 2>&1 ; ls -R /tmp/prof | head -20; cp /tmp/prof/*stats*.csv /tmp/prof/*/*stats*.csv $GRAFT_REPO_ROOT/gpurun_out/ 2>/dev/null; tail -3 /tmp/prof.log )
