#!/bin/bash
# Round-2 GPU pass G: PMC traffic of the GroupNorm / InstanceNorm fused kernels (tools/gn_bench.py under two
# separate --pmc passes), and bench.py with --norm-type gn / in.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/gn_bench.py 2>&1 | grep -v amdgpu > gpurun_out/r02_gn_bench.log; cat gpurun_out/r02_gn_bench.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcg_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcg_$c -o p -- python $GRAFT_REPO_ROOT/tools/gn_bench.py > /tmp/pmcg_$c.log 2>&1
  f=$(find /tmp/pmcg_$c -name '*counter_collection.csv' | head -1)
  (head -1 "$f"; grep -E 'k_gn_|k_bn_res|k_reduce_partials' "$f") > $GRAFT_REPO_ROOT/gpurun_out/r02_pmc_${c}_gn_bench.csv
done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r02_pmc_FETCH_SIZE_gn_bench.csv gpurun_out/r02_pmc_WRITE_SIZE_gn_bench.csv > gpurun_out/r02_pmc_gn_bench.json
grep -A4 "k_gn" gpurun_out/r02_pmc_gn_bench.json | head -60
for nt in gn in; do
  timeout 600 python bench.py --norm-type $nt --no-cpu-baseline --no-stress 2>/dev/null | tail -1 > gpurun_out/r02_bench_cfg_resnet18_$nt.json
  cut -c1-200 gpurun_out/r02_bench_cfg_resnet18_$nt.json
done
