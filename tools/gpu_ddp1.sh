#!/bin/bash
# Single-GPU rehearsal of the multi-GPU launch path (the driver owns the real 8-GPU runs): torch.distributed.run,
# nccl (= RCCL) init, rank-0 state broadcast, FlatSGD's bucketed exchange forced on in a world of one
# (DEEPIPR_FORCE_DDP=1), the hipGraph (forward+backward) + eager exchange form, and the DDP alternative.
export DEEPIPR_FORCE_DDP=1
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
# (with --gpus 1 bench.py would default to graph replay: --eager = what N > 1 ranks run)
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --eager 2>&1 | grep '"metric"' | cut -c1-200
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --graph 2>&1 | grep -E '"metric"|Error' | cut -c1-200
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --scheme 2 --classes 100 --batch 32 --eager 2>&1 | grep '"metric"' | cut -c1-200
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --scheme 2 --classes 100 --batch 32 --graph 2>&1 | grep -E '"metric"|Error' | cut -c1-200
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --ddp --eager 2>&1 | grep '"metric"' | cut -c1-200
