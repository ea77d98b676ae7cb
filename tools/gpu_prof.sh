#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/exp_step.py > gpurun_out/exp_step.log 2>&1; cat gpurun_out/exp_step.log | grep -v amdgpu.ids
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > /tmp/prof.log 2>&1 ; python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r01_kernel_trace.csv --steps 20 --top 40 > $GRAFT_REPO_ROOT/gpurun_out/r01_steady_state.md; tail -1 /tmp/prof.log | cut -c1-200 )
head -60 gpurun_out/r01_steady_state.md
