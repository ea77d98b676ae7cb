#!/bin/bash
# Round-6 PMC evidence: HBM-side traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes with --kernel-trace only) of the
# hand-written kernels INSIDE the eager ResNet50 step at ImageNet geometry (BASELINE config 5, batch 256) and inside the eager
# config-R step.  Summaries: gpurun_out/r06_pmc_in_situ_r50.json / r06_pmc_in_situ_R.json (tools/pmc_in_situ.py conventions:
# bytes per launch, FETCH doubled as the microarchitecture guide prescribes for gfx950).
mkdir -p gpurun_out
export TMPDIR=/tmp
KEEP='k_bn_res|k_bn_dual|k_gamma_beta|k_rank2|k_sgd|k_bn_affine|k_bn_walk|k_gn_|k_conv|k_wino|k_maxpool|k_subsample|k_upsample'
run() {   # tag, bench args...
  tag=$1; shift
  CMD="python $GRAFT_REPO_ROOT/bench.py --eager --no-cpu-baseline --no-stress --no-configs $*"
  (cd $GRAFT_REPO_ROOT && $CMD > gpurun_out/r06_bench_eager_for_pmc_$tag.log 2>&1; grep '"metric"' gpurun_out/r06_bench_eager_for_pmc_$tag.log > gpurun_out/r06_bench_eager_for_pmc_$tag.json)
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${tag}_$c
    timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$c -o p -- $CMD --no-kernel-timing > /tmp/pmc_${tag}_$c.log 2>&1
    f=$(find /tmp/pmc_${tag}_$c -name '*counter_collection.csv' | head -1)
    (head -1 "$f"; grep -E "$KEEP" "$f") > $GRAFT_REPO_ROOT/gpurun_out/r06_pmc_${c}_in_situ_$tag.csv
  done
  cd $GRAFT_REPO_ROOT
  python tools/pmc_in_situ.py gpurun_out/r06_pmc_FETCH_SIZE_in_situ_$tag.csv gpurun_out/r06_pmc_WRITE_SIZE_in_situ_$tag.csv gpurun_out/r06_bench_eager_for_pmc_$tag.json > gpurun_out/r06_pmc_in_situ_$tag.json
  python - $tag <<'PY'
import json, sys
d = json.load(open('gpurun_out/r06_pmc_in_situ_%s.json' % sys.argv[1]))
for k in sorted(d):
    print(k, {kk: d[k][kk] for kk in ('fetch', 'write', 'total', 'algorithmic', 'traffic_over_algorithmic', 'dispatches_fetch_pass') if kk in d[k]})
PY
}
# PMC_CONFIGS: which steps to measure (default both; the ResNet50 passes take ~10 minutes each)
case " ${PMC_CONFIGS:-r50 R} " in *" r50 "*) run r50 --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 3 --warmup 1;; esac
case " ${PMC_CONFIGS:-r50 R} " in *" R "*) run R --steps 12 --warmup 4;; esac
# the raw per-dispatch CSVs are large: keep the summaries and the configuration-R CSVs only
rm -f gpurun_out/r06_pmc_*_in_situ_r50.csv
