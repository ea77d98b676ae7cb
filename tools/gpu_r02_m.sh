#!/bin/bash
# Round-2 GPU pass M: the whole parity suite + smoke on the round's final tree (what the driver runs at round end).
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -60 > gpurun_out/r02_pytest_gpu_final.log
tail -4 gpurun_out/r02_pytest_gpu_final.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
