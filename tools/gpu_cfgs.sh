#!/bin/bash
python bench.py --arch alexnet --batch 64 --steps 100 --warmup 20 --no-stress 2>&1 | grep '"metric"' > gpurun_out/bench_alexnet.json; cut -c1-900 gpurun_out/bench_alexnet.json
python bench.py --image-size 224 --classes 1000 --batch 128 --steps 20 --warmup 5 --no-stress --no-cpu-baseline 2>&1 | grep '"metric"' > gpurun_out/bench_imagenet.json; cut -c1-1500 gpurun_out/bench_imagenet.json
