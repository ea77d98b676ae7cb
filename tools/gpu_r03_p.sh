#!/bin/bash
# Round 3, evidence of the final tree: full -m gpu suite (fresh box), rehearsal table, PMC passes, kernel trace, bench.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_gpu_5.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu_5.log
tail -4 gpurun_out/r03_pytest_gpu_5.log
ROUND_TAG=r03 timeout 900 bash tools/gpu_ddp1.sh
ROUND_TAG=r03 timeout 1200 bash tools/gpu_pmc_in_situ.sh
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-stress > /tmp/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_summary.py $(find /tmp/prof -name '*kernel_trace.csv' | head -1) --steps 40 --top 60 > gpurun_out/r03_steady_state.md
head -61 $(find /tmp/prof -name '*kernel_stats.csv' | head -1) | cut -c1-220 > gpurun_out/r03_rocprofv3_kernel_stats_top60.csv
grep '"metric"' /tmp/prof.log > gpurun_out/r03_bench_under_rocprof.json
head -30 gpurun_out/r03_steady_state.md | cut -c1-140
timeout 600 python bench.py > gpurun_out/r03_bench_n1_final.json 2> gpurun_out/r03_bench_n1_final.err
tail -c 1200 gpurun_out/r03_bench_n1_final.json
