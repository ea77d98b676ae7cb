#!/bin/bash
# Winograd forward / backward-data kernels on the GPU box: parity tests, then per-shape timing against the direct kernel and
# the vendor library.   gpurun --timeout 1500 -- 'bash tools/gpu_wino.sh [notests]'
mkdir -p gpurun_out
if [ "$1" != "notests" ]; then
timeout 1200 python -m pytest tests/test_conv_wino_gpu.py -q -x -p no:cacheprovider 2>&1 | tail -25 > gpurun_out/wino_tests.log
tail -5 gpurun_out/wino_tests.log
fi
for b in 128 32; do
    timeout 300 python tools/conv_bench.py --batch $b --algo winograd --json gpurun_out/conv_bench_winograd_bs$b.json 2>&1 | grep "^{" | head -4 | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['N'], r['Ci'], r['HW'], 'fwd', r['fwd_us'], 'dgrad', r['dgrad_us'], 'lib', r['fwd_us_lib'], r['dgrad_us_lib'], 'err %.1e %.1e' % (r['fwd_err'], r['dgrad_err']))"
done
DEEPIPR_LIB=$GRAFT_REPO_ROOT/deepipr_amd/csrc/libdeepipr_hip_trace.so timeout 300 python tools/wino_trace.py 128 2>&1 | grep -v amdgpu | tee gpurun_out/wino_trace.log
