#!/bin/bash
# The N > 1 code path of bench.py on ONE GPU (the driver owns the multi-GPU node):
#   (1) RCCL world of one with the exchange forced on (DEEPIPR_FORCE_DDP=1): staged graphs + bucket all-reduces + fused SGD
#   (2) TWO gloo ranks sharing the GPU (DEEPIPR_SHARE_GPU=1): rank-0-first find phase, probe-batch broadcast, gradient
#       agreement across ranks, rccl_ranks_seen, the launch-form fallbacks -- functional only, the timing means nothing
mkdir -p gpurun_out
R=${ROUND_TAG:-r04}
OUT=gpurun_out/${R}_nrank_rehearsal.jsonl
: > $OUT
COMMON="--steps 40 --warmup 10 --no-cpu-baseline --no-stress --no-kernel-timing"
DEEPIPR_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 $COMMON 2> gpurun_out/${R}_nrank_1.err | grep '"metric"' | sed 's/^{/{"rehearsal": "nccl world of one, exchange forced on", /' >> $OUT
DEEPIPR_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --backend gloo $COMMON 2> gpurun_out/${R}_nrank_2.err | grep '"metric"' | sed 's/^{/{"rehearsal": "two gloo ranks sharing one GPU", /' >> $OUT
DEEPIPR_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --backend gloo --scheme 3 --classes 100 --batch 64 $COMMON 2> gpurun_out/${R}_nrank_3.err | grep '"metric"' | sed 's/^{/{"rehearsal": "two gloo ranks sharing one GPU, V3", /' >> $OUT
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d['rehearsal'], '|', d['value'], d['ms_per_step'], {k: d.get(k) for k in ('n_gpus', 'world_size_seen', 'rccl_ranks_seen', 'ranks_agree_bitwise', 'ranks_agree_1e-5', 'find_phase_s', 'exchange_timeouts')}, d['config']['launch'][:80])
PY
tail -4 gpurun_out/${R}_nrank_2.err | cut -c1-200
