#!/bin/bash
# Round-2 GPU pass D: final kernels (granule exchange, C=128 split, fwd without table barrier, residual preload).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/${R}_pytest_gpu_d.log
tail -5 gpurun_out/${R}_pytest_gpu_d.log
timeout 600 python tools/res_tune.py > gpurun_out/${R}_res_tune_d.log 2>&1; tail -3 gpurun_out/${R}_res_tune_d.log
bash tools/gpu_pmc_in_situ.sh
timeout 600 python bench.py > gpurun_out/${R}_bench_d.log 2>&1; tail -1 gpurun_out/${R}_bench_d.log | cut -c1-300
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -12 gpurun_out/${R}_steady_state.md | cut -c1-160
bash tools/gpu_cfgs.sh
