#!/bin/bash
# dual form (two norm layers + tail in one launch): kernel and model parity, then A/B in the bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_round3_gpu.py -m gpu -q -p no:cacheprovider -k "dual or shared_trunk" 2>&1 | tail -12
COMMON="--steps 80 --warmup 20 --no-cpu-baseline --no-stress"
one() { python bench.py $COMMON "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print(d['ms_per_step'], d['roofline']['frac'], 'bwd', k['bn_res_bwd']['launches_per_step'], k['bn_res_bwd']['us_per_step'], 'fwd', k['bn_res_fwd']['launches_per_step'], k['bn_res_fwd']['us_per_step'], k['bn_res_fwd']['frac'])"; }
for rep in 1 2 3; do
  echo "V1 dual   $(one)"
  echo "V1 separ. $(DEEPIPR_NO_DUAL_TAIL=1 one)"
done
echo "V2 bs32 dual   $(one --scheme 2 --classes 100 --batch 32)"
echo "V2 bs32 separ. $(DEEPIPR_NO_DUAL_TAIL=1 one --scheme 2 --classes 100 --batch 32)"
