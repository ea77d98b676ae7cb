#!/bin/bash
export TMPDIR=/tmp DEEPIPR_FORCE_DDP=1
cd /tmp && rm -rf /tmp/pd && rocprofv3 --kernel-trace --output-format csv -d /tmp/pd -o d -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-kernel-timing > /tmp/pd.log 2>&1
ls /tmp/pd /tmp/pd/* | head
f=$(ls /tmp/pd/*kernel_trace.csv /tmp/pd/*/*kernel_trace.csv 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/trace_summary.py $f --steps 20 --top 14 | cut -c1-150
grep -i -E "nccl|rccl|AllReduce|ncclDev" $f | awk -F, '{print $8}' | sort | uniq -c | head
