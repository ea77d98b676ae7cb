#!/bin/bash
# A/B of one environment switch on the bench configurations (graph replay, no extras).
#   tools/gpu_ab.sh VAR [reps]      e.g. tools/gpu_ab.sh DEEPIPR_OWN_WGRAD=0
sw=$1; reps=${2:-2}
COMMON="--steps 80 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing"
one() { python bench.py $COMMON "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"; }
for rep in $(seq $reps); do
  echo "R  (V1 bs128)        default $(one)   $sw $(env $sw python bench.py $COMMON 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")"
  echo "P  (V2 c100 bs32)    default $(one --scheme 2 --classes 100 --batch 32)   $sw $(env $sw python bench.py $COMMON --scheme 2 --classes 100 --batch 32 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")"
done
echo "A  (alexnet V1 bs64) default $(one --arch alexnet --batch 64)   $sw $(env $sw python bench.py $COMMON --arch alexnet --batch 64 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")"
