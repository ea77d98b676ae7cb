#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/exp_step.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_step.log | tail -12
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/kbench.py "R " S1 S3 > /tmp/pmc_$c.log 2>&1
  ls /tmp/pmc_$c | head
  cp /tmp/pmc_$c/p_counter_collection.csv $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.csv 2>/dev/null
done
tail -3 /tmp/pmc_WRITE_SIZE.log
