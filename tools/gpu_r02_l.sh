#!/bin/bash
# Round-2 GPU pass L: reworked cross-entropy head: the tests that touch it, bench -> JSON, kernel trace, rehearsal.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -k "cross_entropy or golden or full_size or graph or entry_points or smoke" 2>&1 | tail -30 > gpurun_out/${R}_pytest_gpu_l.log
tail -4 gpurun_out/${R}_pytest_gpu_l.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${R}_bench_n1.json; cut -c1-260 gpurun_out/${R}_bench_n1.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 --marker k_ce_finish > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/${R}_steady_state.md
grep -E "k_ce|k_sgd" gpurun_out/${R}_steady_state.md | cut -c1-40,95-150
ROUND_TAG=$R bash tools/gpu_ddp1.sh
