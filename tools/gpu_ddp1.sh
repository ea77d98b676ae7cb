#!/bin/bash
# Single-GPU rehearsal of the multi-GPU launch path: torch.distributed.run, nccl (RCCL) init, rank-0 state
# broadcast, DistributedDataParallel wrapping and its gradient hooks around the HIP autograd Functions.
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress 2>&1 | grep '"metric"' | cut -c1-200
export DEEPIPR_FORCE_DDP=1
for sg in 0 1; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --ddp-static-graph $sg 2>&1 | grep '"metric"' | cut -c1-200
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --scheme 2 --classes 100 --batch 32 2>&1 | grep '"metric"' | cut -c1-200
