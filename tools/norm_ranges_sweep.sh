#!/bin/bash
# A/B of the norm kernels' range forms on ResNet50's maps: one launch per range | all ranges in one launch, late start 0 .. 8
out=${1:-gpurun_out/norm_ranges_sweep.jsonl}
: > $out
DEEPIPR_BN_RANGES=0 python tools/norm_ranges_bench.py | tail -1 >> $out
for s in 0,0 1,1 2,2 3,3 4,4 6,6 8,8; do
    DEEPIPR_BN_RANGES=1 DEEPIPR_BN_STAGGER=$s python tools/norm_ranges_bench.py | tail -1 >> $out
done
cat $out
