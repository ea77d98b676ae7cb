#!/bin/bash
# Round-2 GPU pass A: full parity suite, knob sweep of the single-pass kernels, bench (default), multi-GPU launch
# path rehearsed on one GPU (exchange forced on in a world of one).
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -80 > gpurun_out/r02_pytest_gpu.log
tail -4 gpurun_out/r02_pytest_gpu.log
timeout 900 python tools/res_tune.py > gpurun_out/r02_res_tune.log 2>&1; tail -3 gpurun_out/r02_res_tune.log
timeout 600 python bench.py > gpurun_out/r02_bench_a.log 2>&1; tail -1 gpurun_out/r02_bench_a.log | cut -c1-400
export DEEPIPR_FORCE_DDP=1
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
for mode in "" "--eager"; do
  $RUN bench.py --gpus 1 --steps 60 --warmup 15 --no-cpu-baseline --no-stress $mode 2>&1 | grep -E '"metric"|Error' | cut -c1-260 >> gpurun_out/r02_ddp1.log
  $RUN bench.py --gpus 1 --steps 60 --warmup 15 --no-cpu-baseline --no-stress --scheme 2 --classes 100 --batch 32 $mode 2>&1 | grep -E '"metric"|Error' | cut -c1-260 >> gpurun_out/r02_ddp1.log
done
unset DEEPIPR_FORCE_DDP
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-stress --scheme 2 --classes 100 --batch 32 2>&1 | grep -E '"metric"|Error' | cut -c1-260 >> gpurun_out/r02_ddp1.log
cat gpurun_out/r02_ddp1.log
