#!/bin/bash
# eager vs hipGraph replay of the default bench configuration (and the V2 small-batch one)
mkdir -p gpurun_out; rm -f gpurun_out/eg.log
for extra in "" "--graph" "--scheme 2 --classes 100 --batch 32" "--scheme 2 --classes 100 --batch 32 --graph"; do
  echo "== $extra" >> gpurun_out/eg.log
  timeout 600 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-stress --no-kernel-timing $extra 2>/dev/null | tail -1 | cut -c1-220 >> gpurun_out/eg.log
done
cat gpurun_out/eg.log
