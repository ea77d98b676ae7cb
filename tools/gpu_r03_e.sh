#!/bin/bash
# round 3, pass E: the round's evidence on the final kernels -- full parity suite, headline bench, steady-state rocprofv3
# profile, in-situ PMC traffic, one-rank RCCL rehearsal (staged / unstaged / eager), GEMV microbench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp ROUND_TAG=r03
O=gpurun_out
rm -f $O/session_end_determinism.json
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r03_pytest_gpu_1.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest_gpu_1.log; tail -4 $O/r03_pytest_gpu_1.log
timeout 300 python __graft_entry__.py smoke > $O/r03_smoke.log 2>&1; tail -1 $O/r03_smoke.log
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/r03_bench_n1.json; cut -c1-400 $O/r03_bench_n1.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r03 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r03_kernel_trace.csv --steps 40 --top 45 --gaps 8 > $GRAFT_REPO_ROOT/$O/r03_steady_state.md
head -60 /tmp/prof/r03_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/$O/r03_rocprofv3_kernel_stats_top60.csv
grep '"metric"' /tmp/prof.log | cut -c1-4000 > $GRAFT_REPO_ROOT/$O/r03_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -3 $O/r03_steady_state.md; grep "k_" $O/r03_steady_state.md | cut -c1-150 | head -16
python tools/gemv_bench.py > $O/r03_gemv_bench.json 2>/dev/null; python tools/gemv_bench.py --flush > $O/r03_gemv_bench_flush.json 2>/dev/null
bash tools/gpu_pmc_in_situ.sh
bash tools/gpu_ddp1.sh
