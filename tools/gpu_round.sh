#!/bin/bash
# GPU round script: parity tests, smoke, bench, rocprofv3 kernel trace (+stats), kernel micro-bench, PMC passes.
# Run via gpurun from the repo root; results land in gpurun_out/ and are copied into profiles/ by hand.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r01_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/r01_steady_state.md
head -60 /tmp/prof/r01_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/r01_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-2500 > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.json
rm -rf /tmp/kb && rocprofv3 --kernel-trace --output-format csv -d /tmp/kb -o kb -- python $GRAFT_REPO_ROOT/tools/kbench.py > $GRAFT_REPO_ROOT/gpurun_out/kbench.log 2>&1; cp /tmp/kb/kb_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py "R " S3 > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT; grep -E "wall per" gpurun_out/r01_steady_state.md
