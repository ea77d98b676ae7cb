#!/bin/bash
# Full `-m gpu` session on the GPU box; the log lands in gpurun_out/ (copy the one to be judged into profiles/).
#   tools/gpu_suite.sh [tag] [extra pytest args...]
tag=${1:-suite}; shift
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" > gpurun_out/pytest_gpu_$tag.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/pytest_gpu_$tag.log | cut -c1-250 | tail -30
exit $rc
