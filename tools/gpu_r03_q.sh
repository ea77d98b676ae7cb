#!/bin/bash
# Round 3, last evidence call: full -m gpu suite on the final tree, two gloo ranks sharing the GPU through the
# self-launcher (V1 and V2: staged step, launch-form agreement, shared trunk), the other configurations' bench lines.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r03_pytest_gpu_6.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu_6.log
grep -E "passed|failed|^FAILED" gpurun_out/r03_pytest_gpu_6.log | tail -5
: > gpurun_out/r03_two_ranks_one_gpu.jsonl
for cfg in "" "--scheme 2 --classes 100 --batch 32"; do
  DEEPIPR_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-stress --no-kernel-timing $cfg \
      2> gpurun_out/r03_two_ranks.err | grep '"metric"' >> gpurun_out/r03_two_ranks_one_gpu.jsonl
  echo "two ranks rc=$? $cfg"; tail -2 gpurun_out/r03_two_ranks.err | cut -c1-200
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_two_ranks_one_gpu.jsonl'):
    d = json.loads(l)
    print(d['n_gpus'], d.get('world_size_seen'), d['value'], d['ms_per_step'], d.get('exchange_us_exposed'), d.get('exchange_timeouts'), d['config']['launch'][:60])
PY
run() { out=$1; shift; timeout 900 python bench.py "$@" 2>/dev/null | grep '"metric"' > gpurun_out/$out; python - "$out" <<'PY'
import json, sys
d = json.load(open('gpurun_out/' + sys.argv[1]))
r = d.get('roofline', {})
print(sys.argv[1], d['value'], 'img/s', d['ms_per_step'], 'ms', '|', r.get('kernel', '')[:14], r.get('frac'), r.get('avg_us'), r.get('launches_per_step'))
PY
}
run r03_bench_cfg_alexnet.json --arch alexnet --batch 64 --steps 100 --warmup 20 --no-stress --no-cpu-baseline
run r03_bench_cfg_alexnet_v2.json --arch alexnet --scheme 2 --batch 64 --steps 100 --warmup 20 --no-stress --no-cpu-baseline
run r03_bench_cfg_v2_bs128.json --scheme 2 --classes 100 --batch 128 --steps 60 --warmup 15 --no-stress --no-cpu-baseline
run r03_bench_cfg_resnet18_gn.json --norm-type gn --steps 60 --warmup 15 --no-stress --no-cpu-baseline
