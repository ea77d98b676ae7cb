#!/bin/bash
# Single-GPU rehearsal of the multi-GPU launch path (the driver owns the real 8-GPU runs): torch.distributed.run,
# nccl (= RCCL) init, rank-0 state broadcast, FlatSGD's bucketed exchange forced on in a world of one
# (DEEPIPR_FORCE_DDP=1), and the DistributedDataParallel alternative.
export DEEPIPR_FORCE_DDP=1
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress 2>&1 | grep '"metric"' | cut -c1-200
$RUN bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-stress --ddp 2>&1 | grep '"metric"' | cut -c1-200
$RUN train_v1.py --arch resnet --train-passport --key-type random --epochs 1 --passport-config passport_configs/resnet18_passport.json --batch-size 128 --synthetic-samples 1280 --logdir /tmp/ddp_logs 2>&1 | grep -E "epoch 1|Error|error" | cut -c1-300
$RUN train_v23.py --arch resnet --key-type random --epochs 1 --passport-config passport_configs/resnet18_passport.json --batch-size 64 --synthetic-samples 640 --dataset cifar100 --logdir /tmp/ddp_logs 2>&1 | grep -E "epoch 1|Error|error" | cut -c1-300
