#!/bin/bash
# Round-5 PMC evidence (separate rocprofv3 --pmc passes, --kernel-trace only):
#   (1) in-situ HBM traffic of the hand-written kernels inside the eager config-R step (FETCH_SIZE / WRITE_SIZE), conv kernels included
#   (2) matrix-core counters of the convolution kernels under tools/wgrad_bench.py / tools/conv_bench.py
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${ROUND_TAG:-r05}
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES\b" | cut -c1-160 | sort -u | head -30 > gpurun_out/${R}_pmc_counters_available.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --eager --steps 12 --warmup 4 --no-cpu-baseline --no-stress"
$CMD > gpurun_out/${R}_bench_eager_for_pmc.log 2>&1
grep '"metric"' gpurun_out/${R}_bench_eager_for_pmc.log > gpurun_out/${R}_bench_eager_for_pmc.json
cd /tmp
KEEP='k_bn_res|k_bn_dual|k_gamma_beta|k_rank2|k_sgd|k_bn_affine|k_bn_walk|k_gn_|k_conv|k_wino'
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- $CMD --no-kernel-timing > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  (head -1 "$f"; grep -E "$KEEP" "$f") > $GRAFT_REPO_ROOT/gpurun_out/${R}_pmc_${c}_in_situ.csv
done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python $GRAFT_REPO_ROOT/tools/wgrad_bench.py --reps 4 --no-find > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
  (head -1 "$f"; grep -E "k_conv" "$f") > $GRAFT_REPO_ROOT/gpurun_out/${R}_pmc_${c}_wgrad_bench.csv
done
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc2_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc2_$c -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --reps 4 > /tmp/pmc2_$c.log 2>&1
  f=$(find /tmp/pmc2_$c -name '*counter_collection.csv' | head -1)
  (head -1 "$f"; grep -E "k_conv" "$f") > $GRAFT_REPO_ROOT/gpurun_out/${R}_pmc_${c}_conv_bench.csv
done
cd $GRAFT_REPO_ROOT
python tools/pmc_in_situ.py gpurun_out/${R}_pmc_FETCH_SIZE_in_situ.csv gpurun_out/${R}_pmc_WRITE_SIZE_in_situ.csv gpurun_out/${R}_bench_eager_for_pmc.json > gpurun_out/${R}_pmc_in_situ.json
python - <<'PY'
import collections, csv, json, os
R = os.environ.get('ROUND_TAG', 'r05')
d = json.load(open('gpurun_out/%s_pmc_in_situ.json' % R))
for k in sorted(d):
    print(k, {kk: d[k][kk] for kk in ('fetch', 'write', 'total', 'algorithmic', 'traffic_over_algorithmic') if kk in d[k]})
acc = {}
for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'SQ_BUSY_CYCLES'):
    per = collections.defaultdict(lambda: [0.0, 0])
    try:
        for row in csv.DictReader(open('gpurun_out/%s_pmc_%s_wgrad_bench.csv' % (R, c))):
            name = row['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:48]
            per[name][0] += float(row['Counter_Value']); per[name][1] += 1
    except OSError:
        continue
    acc[c] = {k: v[0] / v[1] for k, v in per.items()}
for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'):
    per = collections.defaultdict(lambda: [0.0, 0])
    try:
        for row in csv.DictReader(open('gpurun_out/%s_pmc_%s_conv_bench.csv' % (R, c))):
            name = row['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:48]
            per[name][0] += float(row['Counter_Value']); per[name][1] += 1
    except OSError:
        continue
    acc.setdefault(c, {}).update({k: v[0] / v[1] for k, v in per.items()})
json.dump(acc, open('gpurun_out/%s_pmc_mfma_wgrad_bench.json' % R, 'w'), indent=1)
for k in sorted(acc.get('SQ_VALU_MFMA_BUSY_CYCLES', {})):
    print(k, {c: round(acc[c].get(k, 0)) for c in acc})
PY
cat gpurun_out/${R}_pmc_counters_available.txt | head -12
