#!/bin/bash
# In-situ PMC traffic of the hand-written kernels inside the config-R train step: two separate rocprofv3 passes
# (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as gpurun requires) over the eager bench.py step, plus the same
# command un-profiled for the algorithmic bytes.  -> gpurun_out/${R}_pmc_in_situ.json (+ the trimmed counter CSVs)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND_TAG:-r02}
CMD="python $GRAFT_REPO_ROOT/bench.py --eager --steps 12 --warmup 4 --no-cpu-baseline --no-stress"
$CMD > gpurun_out/${R}_bench_eager_for_pmc.log 2>&1
grep '"metric"' gpurun_out/${R}_bench_eager_for_pmc.log > gpurun_out/${R}_bench_eager_for_pmc.json
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- $CMD --no-kernel-timing > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  # keep the hand-written kernels' rows only (the raw file has every MIOpen / ATen dispatch)
  (head -1 "$f"; grep -E 'k_bn_res|k_bn_dual|k_gamma_beta|k_rank2|k_sgd|k_bn_affine|k_bn_walk|k_gn_' "$f") > $GRAFT_REPO_ROOT/gpurun_out/${R}_pmc_${c}_in_situ.csv
done
cd $GRAFT_REPO_ROOT
python tools/pmc_in_situ.py gpurun_out/${R}_pmc_FETCH_SIZE_in_situ.csv gpurun_out/${R}_pmc_WRITE_SIZE_in_situ.csv gpurun_out/${R}_bench_eager_for_pmc.json > gpurun_out/${R}_pmc_in_situ.json
python - <<'PY'
import json, os
d = json.load(open('gpurun_out/%s_pmc_in_situ.json' % os.environ.get('ROUND_TAG', 'r02')))
for k in ('k_bn_res_bwd', 'k_bn_res_fwd', 'k_sgd', 'k_gamma_beta_multi', 'k_rank2_multi'):
    if k in d:
        print(k, d[k])
PY
