#!/bin/bash
# Round-2 GPU pass B: full parity suite (conv-inside node, NaN-propagating ReLU), phase trace of the single-pass
# kernels, in-situ PMC traffic, bench + rocprofv3 kernel trace of the same command.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/${R}_pytest_gpu_b.log
tail -5 gpurun_out/${R}_pytest_gpu_b.log
timeout 600 python tools/res_trace.py > gpurun_out/${R}_res_trace.log 2>&1; tail -12 gpurun_out/${R}_res_trace.log
bash tools/gpu_pmc_in_situ.sh
timeout 600 python bench.py > gpurun_out/${R}_bench_b.log 2>&1; tail -1 gpurun_out/${R}_bench_b.log | cut -c1-300
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -12 gpurun_out/${R}_steady_state.md
