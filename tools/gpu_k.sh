#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
( cd /tmp && rm -rf /tmp/kb && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kb -o kb -- python $GRAFT_REPO_ROOT/tools/kbench.py > $GRAFT_REPO_ROOT/gpurun_out/kbench.log 2>&1; grep -E "k_affine|k_gamma|k_passport|k_reduce|elementwise" /tmp/kb/kb_kernel_stats.csv | cut -c1-150,400- > $GRAFT_REPO_ROOT/gpurun_out/kbench_rocprof.csv; cp /tmp/kb/kb_kernel_trace.csv $GRAFT_REPO_ROOT/gpurun_out/ )
grep -v amdgpu.ids gpurun_out/kbench.log
