#!/bin/bash
# Round-2 GPU pass H: final tree after the launch-count trims (SignLoss adds, stem fork): parity suite, bench,
# rocprofv3 kernel trace + stats of the same command.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -120 > gpurun_out/${R}_pytest_gpu_h.log
tail -4 gpurun_out/${R}_pytest_gpu_h.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 > gpurun_out/${R}_bench_h.json; cut -c1-260 gpurun_out/${R}_bench_h.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state_h.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top_h.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof_h.json
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/${R}_steady_state_h.md
bash tools/gpu_pmc_in_situ.sh
