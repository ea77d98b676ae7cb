#!/bin/bash
# GroupNorm / InstanceNorm nets: fused kernels (deepipr_passport_gn_*) against the library norm + unfused passport kernels
# usage: bash tools/gpu_norms.sh [fused-only]
mkdir -p gpurun_out; rm -f gpurun_out/norms.log
variants=("" "--no-fuse"); [ "$1" = "fused-only" ] && variants=("")
for nt in gn in; do for extra in "${variants[@]}"; do
  echo "== $nt $extra" >> gpurun_out/norms.log
  timeout 600 python bench.py --norm-type $nt $extra --steps 60 --warmup 15 --no-cpu-baseline --no-stress 2>/dev/null | grep '"metric"' > gpurun_out/bench_${nt}${extra}.json
  python - gpurun_out/bench_${nt}${extra}.json >> gpurun_out/norms.log <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d['value'], 'img/s', d['ms_per_step'], 'ms |', {k: (v.get('launches_per_step'), v.get('avg_us'), v.get('frac')) for k, v in d.get('kernels', {}).items()})
PY
done; done
cat gpurun_out/norms.log
