#!/bin/bash
# Round-2 GPU pass E (evidence run on the final tree): smoke, the new adversarial-rows test, bench.py (default) ->
# JSON line, rocprofv3 kernel trace + stats of the same command, the other BASELINE configurations.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=r02
timeout 300 python __graft_entry__.py smoke > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log
timeout 900 python -m pytest tests/test_round2_gpu.py -m gpu -q -p no:cacheprovider -k "adversarial or exchange or statistics" 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/${R}_bench_e.log 2>&1; tail -1 gpurun_out/${R}_bench_e.log > gpurun_out/${R}_bench_n1.json; cut -c1-200 gpurun_out/${R}_bench_n1.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/${R}_steady_state.md
timeout 300 python tools/res_tune.py 2>&1 | grep -E "plain|tail|default" > gpurun_out/${R}_res_tune_e.log; tail -4 gpurun_out/${R}_res_tune_e.log
bash tools/gpu_cfgs.sh
