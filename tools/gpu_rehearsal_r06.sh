#!/bin/bash
# Round-6 rehearsals of the multi-GPU path on ONE GPU (the driver owns the real 8-GPU node): gpurun_out/r06_nrank_rehearsal.jsonl
#   1. config-P shard (V2, 100 classes, 32 images: the configuration with the least room to hide a 44.9 MB all-reduce) in an RCCL
#      world of one with the exchange forced on (DEEPIPR_FORCE_DDP=1), host-wait histogram included, and
#   2. the same shard without any exchange: line 1's step must stay within 1.06 x of it;
#   3. / 4. V3 with two gloo ranks sharing the GPU, MIOpen in find mode (round 5: ranks not bit-identical) and in immediate mode
#      (--no-miopen-find): if the ranks agree bit for bit in immediate mode, the disagreement is the vendor library's per-process
#      solver choice for the shapes this library has no kernel for (batch 66: 4-wide maps need a multiple of four images).
mkdir -p gpurun_out
OUT=gpurun_out/r06_nrank_rehearsal.jsonl
: > $OUT
COMMON="--steps 100 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing --no-configs"
P="--scheme 2 --classes 100 --batch 32"
RUN1="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29521"
DEEPIPR_FORCE_DDP=1 timeout 300 $RUN1 bench.py --gpus 1 $COMMON $P --stage-host-wait-histogram 2>/dev/null | grep -E '"metric"' | sed 's/^{/{"rehearsal": "config-P shard, nccl world of one, exchange forced on", /' >> $OUT
timeout 300 python bench.py --gpus 1 $COMMON $P 2>/dev/null | grep -E '"metric"' | sed 's/^{/{"rehearsal": "config-P shard, one GPU, no exchange", /' >> $OUT
RUN2="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29522"
V3="--scheme 3 --classes 100 --batch 64 --steps 30 --warmup 8 --no-cpu-baseline --no-stress --no-kernel-timing --no-configs --backend gloo"
DEEPIPR_SHARE_GPU=1 timeout 400 $RUN2 bench.py --gpus 2 $V3 2>/dev/null | grep -E '"metric"' | sed 's/^{/{"rehearsal": "V3, two gloo ranks sharing one GPU, MIOpen find mode", /' >> $OUT
DEEPIPR_SHARE_GPU=1 timeout 400 $RUN2 bench.py --gpus 2 $V3 --no-miopen-find 2>/dev/null | grep -E '"metric"' | sed 's/^{/{"rehearsal": "V3, two gloo ranks sharing one GPU, MIOpen immediate mode", /' >> $OUT
python - "$OUT" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    sp = (d.get('config') or {}).get('exchange') or {}
    print('%-62s %8.1f img/s %7.3f ms  exposed %s us  agree_bitwise %s  backend %s  host_wait %s' % (
        d['rehearsal'], d['value'], d['ms_per_step'], d.get('exchange_us_exposed'), d.get('ranks_agree_bitwise'),
        d.get('ranks_seen_backend'), json.dumps(sp.get('host_wait_us'))))
PY
