#!/bin/bash
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r01_kernel_trace.csv --steps 20 --top 40 > $GRAFT_REPO_ROOT/gpurun_out/r01_steady_state.md
cut -c1-150 $GRAFT_REPO_ROOT/gpurun_out/r01_steady_state.md | head -60
