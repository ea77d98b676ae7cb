#!/bin/bash
export DEEPIPR_FORCE_DDP=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 60 --warmup 15 --no-cpu-baseline --no-stress 2>&1 | grep -E '"metric"|Error|error' | cut -c1-170
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 60 --warmup 15 --no-cpu-baseline --no-stress --scheme 2 --classes 100 --batch 32 2>&1 | grep -E '"metric"|Error|error' | cut -c1-170
