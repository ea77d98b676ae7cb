#!/bin/bash
# GPU round script: parity tests, smoke, bench, rocprofv3 kernel trace (+stats), per-shape kernel bench, PMC passes.
# Run via gpurun from the repo root; results land in gpurun_out/ and the ones to be judged are copied into profiles/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND_TAG:-r01}
if [ -z "$SKIP_TESTS" ]; then
  timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
  tail -3 gpurun_out/pytest_gpu.log
fi
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log | cut -c1-300
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/bench_under_rocprof.json
timeout 300 python $GRAFT_REPO_ROOT/tools/res_bench.py > $GRAFT_REPO_ROOT/gpurun_out/res_bench.log 2>&1
rm -rf /tmp/kb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -o kb -- python $GRAFT_REPO_ROOT/tools/res_bench.py > /tmp/kb.log 2>&1
head -40 /tmp/kb/kb_kernel_stats.csv | cut -c1-300 > $GRAFT_REPO_ROOT/gpurun_out/${R}_res_bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/res_bench.py > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv > gpurun_out/pmc_per_kernel.json
grep -E "wall per" gpurun_out/${R}_steady_state.md
