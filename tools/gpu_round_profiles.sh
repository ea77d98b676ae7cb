#!/bin/bash
# The round's measurement set: steady-state rocprofv3 summaries (config R, config P shard), the ImageNet-shape lines.
#   tools/gpu_round_profiles.sh <round tag, e.g. r04>
tag=${1:-rXX}
tools/gpu_steady.sh ${tag}_steady_state > /dev/null 2>&1
head -12 gpurun_out/${tag}_steady_state.md | cut -c1-150
MARKERS_ARGS="--markers-per-step 2" tools/gpu_steady.sh ${tag}_steady_state_cfg_P --scheme 2 --classes 100 --batch 32 > /dev/null 2>&1
head -8 gpurun_out/${tag}_steady_state_cfg_P.md | cut -c1-150
COMMON="--steps 20 --warmup 5 --no-cpu-baseline --no-stress"
timeout 900 python bench.py $COMMON --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find > gpurun_out/${tag}_bench_r50_bs256.json 2> gpurun_out/${tag}_bench_r50_bs256.err; echo "r50 bs256 rc=$?"
timeout 600 python bench.py $COMMON --arch resnet50 --image-size 224 --classes 1000 --batch 64 > gpurun_out/${tag}_bench_r50_bs64.json 2> /dev/null; echo "r50 bs64 rc=$?"
timeout 600 python bench.py $COMMON --image-size 224 --classes 1000 --batch 128 > gpurun_out/${tag}_bench_r18_224.json 2> /dev/null; echo "r18 224 rc=$?"
for f in r50_bs256 r50_bs64 r18_224; do python -c "
import json; d=json.loads(open('gpurun_out/${tag}_bench_$f.json').read().strip().splitlines()[-1]); r=d.get('roofline_hbm', d['roofline']); print('$f', d['value'], d['ms_per_step'], r['kernel'][:40], r['frac'], r.get('launches_per_step'))"; done
