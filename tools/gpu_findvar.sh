#!/bin/bash
# run-to-run variance of the headline step with MIOpen find mode (cudnn.benchmark=True) and without
mkdir -p gpurun_out; rm -f gpurun_out/findvar.log
for rep in 1 2 3 4; do for extra in "" "--no-miopen-find"; do
  echo -n "rep$rep [$extra] " >> gpurun_out/findvar.log
  timeout 300 python bench.py $extra --steps 60 --warmup 15 --no-cpu-baseline --no-stress --no-kernel-timing 2>/dev/null | grep '"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> gpurun_out/findvar.log
done; done
cat gpurun_out/findvar.log
