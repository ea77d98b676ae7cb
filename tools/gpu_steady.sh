#!/bin/bash
# steady-state kernel summary of the bench step under rocprofv3.   tools/gpu_steady.sh tag [bench args]
tag=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/prof && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-stress "$@" > /tmp/prof.log 2>&1
tail -2 /tmp/prof.log | cut -c1-300
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r_kernel_trace.csv --steps 40 --top 60 $MARKERS_ARGS > $GRAFT_REPO_ROOT/gpurun_out/${tag}.md
cut -c1-160 $GRAFT_REPO_ROOT/gpurun_out/${tag}.md | head -75
