#!/bin/bash
# Single-GPU rehearsal of the multi-GPU launch path (the driver owns the real 8-GPU runs): torch.distributed.run,
# nccl (= RCCL) init, rank-0 state broadcast, FlatSGD's bucketed exchange forced on in a world of one
# (DEEPIPR_FORCE_DDP=1).  For config R (V1, 128 images/GPU) and the config-P shard (V2, 32 images/GPU):
#   default    = what N > 1 ranks run: the staged step (one hipGraph per backward stage, bucket all-reduces in between)
#   --unstaged = round 2's form: one graph of zero_grad..backward, the whole exchange after it
#   --eager  = eager dispatch with the exchange overlapped with backward
#   single   = the plain one-GPU run (whole step in one graph, no exchange): the number the scaling is judged against
# Full JSON lines -> gpurun_out/${R}_ddp_rehearsal.jsonl
mkdir -p gpurun_out
R=${ROUND_TAG:-r03}
OUT=gpurun_out/${R}_ddp_rehearsal.jsonl
: > $OUT
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
COMMON="--gpus 1 --steps 60 --warmup 15 --no-cpu-baseline --no-stress --no-kernel-timing"
for cfg in "" "--scheme 2 --classes 100 --batch 32"; do
  for mode in "" "--unstaged" "--eager"; do
    DEEPIPR_FORCE_DDP=1 $RUN bench.py $COMMON $cfg $mode 2>&1 | grep -E '"metric"' | sed "s/^{/{\"rehearsal\": \"exchange forced on, ${mode:-default (staged graphs, overlapped exchange)}\", /" >> $OUT
  done
  python bench.py $COMMON $cfg 2>&1 | grep -E '"metric"' | sed 's/^{/{"rehearsal": "single GPU, whole step in one graph, no exchange", /' >> $OUT
done
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print('%-60s %-50s %8.1f img/s %7.3f ms' % (d['rehearsal'], d['config']['workload'][:50], d['value'], d['ms_per_step']))
PY
