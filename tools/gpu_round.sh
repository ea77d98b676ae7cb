#!/bin/bash
# One GPU round of evidence, via gpurun from the repo root (results land in gpurun_out/; copy what is to be judged into
# profiles/):   ROUND_TAG=r03 bash tools/gpu_round.sh        [SKIP_TESTS=1 skips the ~10 min parity suite]
#   1. parity suite (-m gpu) + smoke
#   2. bench.py (default)            -> ${R}_bench_n1.json
#   3. rocprofv3 --kernel-trace --stats of the same command -> ${R}_steady_state.md, ${R}_kernel_stats_top.csv
#   4. per-shape kernel bench with the planner knobs (tools/res_tune.py), phase stamps (measurement build)
#   5. in-situ PMC traffic: two separate --pmc passes (tools/gpu_pmc_in_situ.sh)
#   6. multi-GPU launch path rehearsed on one GPU (tools/gpu_ddp1.sh), other BASELINE configurations (tools/gpu_cfgs.sh)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND_TAG:-r03}
export ROUND_TAG=$R
if [ -z "$SKIP_TESTS" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/${R}_pytest_gpu.log
  tail -3 gpurun_out/${R}_pytest_gpu.log
fi
timeout 300 python __graft_entry__.py smoke > gpurun_out/${R}_smoke.log 2>&1; tail -1 gpurun_out/${R}_smoke.log
timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/${R}_bench_n1.json; cut -c1-300 gpurun_out/${R}_bench_n1.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o $R -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/${R}_kernel_trace.csv --steps 40 --top 45 > $GRAFT_REPO_ROOT/gpurun_out/${R}_steady_state.md
head -60 /tmp/prof/${R}_kernel_stats.csv | cut -c1-400 > $GRAFT_REPO_ROOT/gpurun_out/${R}_kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-3500 > $GRAFT_REPO_ROOT/gpurun_out/${R}_bench_under_rocprof.json
cd $GRAFT_REPO_ROOT
head -3 gpurun_out/${R}_steady_state.md
timeout 600 python tools/res_tune.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_res_tune.log
make -C deepipr_amd/csrc trace > /dev/null 2>&1
DEEPIPR_LIB=$GRAFT_REPO_ROOT/deepipr_amd/csrc/libdeepipr_hip_trace.so timeout 600 python tools/res_trace.py 2>&1 | grep -v amdgpu > gpurun_out/${R}_res_trace.log
bash tools/gpu_pmc_in_situ.sh
bash tools/gpu_ddp1.sh
bash tools/gpu_cfgs.sh
