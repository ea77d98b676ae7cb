#!/bin/bash
# GPU round script: parity tests, smoke, bench, kernel trace, PMC.  Run via gpurun from the repo root.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --steps 100 --warmup 20 > gpurun_out/bench.log 2>&1; tail -1 gpurun_out/bench.log
timeout 600 python bench.py --steps 50 --warmup 10 --scheme 2 --classes 100 --batch 32 --no-cpu-baseline --no-stress > gpurun_out/bench_v2.log 2>&1; tail -1 gpurun_out/bench_v2.log | cut -c1-400
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o r01 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r01_kernel_trace.csv --steps 20 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/r01_steady_state.md
rm -rf /tmp/kb && rocprofv3 --kernel-trace --output-format csv -d /tmp/kb -o kb -- python $GRAFT_REPO_ROOT/tools/kbench.py > $GRAFT_REPO_ROOT/gpurun_out/kbench.log 2>&1; cp /tmp/kb/kb_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py "R " S3 > /tmp/pmc_$c.log 2>&1
  cp /tmp/pmc_$c/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.csv 2>/dev/null
done
cd $GRAFT_REPO_ROOT; grep -E "wall per|k_" gpurun_out/r01_steady_state.md | cut -c1-160
