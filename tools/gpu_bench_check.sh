#!/bin/bash
# bench lines of the round: default (config R), V3 (config 4 shard), P shard -> gpurun_out/bench_<tag>_*.json
tag=${1:-x}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/bench_${tag}_R.json 2> gpurun_out/bench_${tag}_R.err; echo "R rc=$?"
python - gpurun_out/bench_${tag}_R.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'rccl_ranks_seen', 'ranks_agree_bitwise', 'find_phase_s')})
for k in ('roofline', 'roofline_hbm', 'roofline_mfma', 'roofline_passport', 'cpu_baseline'):
    if k in d:
        r = dict(d[k]); r.pop('kernels', None); r.pop('note', None)
        print(k, json.dumps(r)[:600])
PY
COMMON="--steps 60 --warmup 15 --no-cpu-baseline --no-stress"
timeout 600 python bench.py $COMMON --scheme 3 --classes 100 --batch 64 > gpurun_out/bench_${tag}_V3.json 2> gpurun_out/bench_${tag}_V3.err; echo "V3 rc=$?"
timeout 600 python bench.py $COMMON --scheme 2 --classes 100 --batch 32 > gpurun_out/bench_${tag}_P.json 2> gpurun_out/bench_${tag}_P.err; echo "P rc=$?"
for c in V3 P; do python -c "
import json,sys; d=json.loads(open('gpurun_out/bench_${tag}_$c.json').read().strip().splitlines()[-1]); print('$c', d['value'], d['ms_per_step'], d['config']['workload'][:120]); print('  passport', {k: v for k, v in d.get('roofline_passport', {}).items() if k in ('frac','us_per_step','achieved')})"; done
