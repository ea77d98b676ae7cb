#!/bin/bash
# bench.py on the other BASELINE configurations (A, P shard eager/graph, I, ResNet50 variant) -> gpurun_out/bench_*.json
mkdir -p gpurun_out
run() { out=$1; shift; timeout 900 python bench.py "$@" 2>&1 | grep '"metric"' > gpurun_out/$out; python - "$out" <<'PY'
import json, sys
d = json.load(open('gpurun_out/' + sys.argv[1]))
r = d.get('roofline', {})
print(sys.argv[1], d['value'], 'img/s', d['ms_per_step'], 'ms', '|', r.get('kernel', '')[:14], r.get('frac'), r.get('avg_us'), r.get('launches_per_step'),
      '|', {k: (v.get('launches_per_step'), v.get('avg_us')) for k, v in d.get('kernels', {}).items()})
PY
}
run bench_alexnet.json --arch alexnet --batch 64 --steps 100 --warmup 20 --no-stress --no-cpu-baseline
run bench_v2_eager.json --scheme 2 --classes 100 --batch 32 --steps 100 --warmup 20 --no-stress --no-cpu-baseline --eager
run bench_v2_graph.json --scheme 2 --classes 100 --batch 32 --steps 100 --warmup 20 --no-stress --no-cpu-baseline
run bench_imagenet.json --image-size 224 --classes 1000 --batch 128 --steps 20 --warmup 5 --no-stress --no-cpu-baseline
run bench_r50.json --arch resnet50 --image-size 224 --classes 1000 --batch 64 --steps 20 --warmup 5 --no-stress --no-cpu-baseline
run bench_r50_bs256.json --arch resnet50 --image-size 224 --classes 1000 --batch 256 --steps 10 --warmup 3 --no-stress --no-cpu-baseline
