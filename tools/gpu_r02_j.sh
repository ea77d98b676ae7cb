#!/bin/bash
# Round-2 GPU pass J: fused cross-entropy head: parity suite, smoke, bench, kernel trace.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/${R}_pytest_gpu_j.log
tail -4 gpurun_out/${R}_pytest_gpu_j.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${R}_bench_j.json; cut -c1-260 gpurun_out/${R}_bench_j.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 60 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state_j.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top_j.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof_j.json
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/${R}_steady_state_j.md
