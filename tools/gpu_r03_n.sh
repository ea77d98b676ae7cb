#!/bin/bash
# shared trunk of the V2 / V3 dual forward: parity, then config P / V3-like numbers with and without it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "shared_trunk or grouped_rank2" 2>&1 | tail -6
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_round2_gpu.py -m gpu -q -x -p no:cacheprovider -k "private or v2 or v3 or V2 or staged or graph" 2>&1 | tail -6
COMMON="--steps 80 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing"
one() { python bench.py $COMMON "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
  echo "V2 bs32  shared $(one --scheme 2 --classes 100 --batch 32)   twice $(DEEPIPR_NO_SHARED_TRUNK=1 one --scheme 2 --classes 100 --batch 32)"
done
echo "V2 bs128 shared $(one --scheme 2 --classes 100 --batch 128)   twice $(DEEPIPR_NO_SHARED_TRUNK=1 one --scheme 2 --classes 100 --batch 128)"
echo "V2 alexnet bs64 shared $(one --arch alexnet --scheme 2 --batch 64)   twice $(DEEPIPR_NO_SHARED_TRUNK=1 one --arch alexnet --scheme 2 --batch 64)"
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513"
echo "V2 bs32 staged rehearsal: $(DEEPIPR_FORCE_DDP=1 $RUN bench.py --gpus 1 $COMMON --scheme 2 --classes 100 --batch 32 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('exchange_us_exposed'))")"
python bench.py --scheme 2 --classes 100 --batch 32 --steps 60 --warmup 15 > gpurun_out/r03_bench_cfg_P_shard.json 2>/dev/null
tail -c 600 gpurun_out/r03_bench_cfg_P_shard.json
