#!/bin/bash
# Round-2 GPU pass C: granule exchange + XCD-aware slice placement; full parity suite, knob sweep, phase trace
# (measurement build), in-situ PMC traffic, bench + rocprofv3 kernel trace, multi-GPU launch path rehearsal.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -150 > gpurun_out/${R}_pytest_gpu_c.log
tail -5 gpurun_out/${R}_pytest_gpu_c.log
timeout 600 python tools/res_tune.py > gpurun_out/${R}_res_tune_c.log 2>&1; tail -3 gpurun_out/${R}_res_tune_c.log
make -C deepipr_amd/csrc trace > /dev/null 2>&1
DEEPIPR_LIB=$GRAFT_REPO_ROOT/deepipr_amd/csrc/libdeepipr_hip_trace.so timeout 600 python tools/res_trace.py > gpurun_out/${R}_res_trace_c.log 2>&1; tail -12 gpurun_out/${R}_res_trace_c.log | cut -c1-300
bash tools/gpu_pmc_in_situ.sh
timeout 600 python bench.py > gpurun_out/${R}_bench_c.log 2>&1; tail -1 gpurun_out/${R}_bench_c.log | cut -c1-300
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -12 gpurun_out/${R}_steady_state.md | cut -c1-160
ROUND_TAG=$R bash tools/gpu_ddp1.sh
