#!/bin/bash
# GPU round script: parity tests, smoke, bench, kernel trace.  Run via gpurun from the repo root.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
( cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /tmp/prof.log 2>&1 ; python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r01_kernel_trace.csv --steps 20 --top 12 > $GRAFT_REPO_ROOT/gpurun_out/r01_steady_state.md; tail -1 /tmp/prof.log | cut -c1-150 )
grep -E "wall per|k_" gpurun_out/r01_steady_state.md
