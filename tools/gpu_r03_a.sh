#!/bin/bash
# round 3, pass A: new GEMV kernels + staged exchange (parity, microbench, one-rank RCCL rehearsal, two-rank gloo
# rehearsal on the one GPU), vendor-convolution determinism probes, kernel list of a pinned-MIOpen step
cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_parity_gpu.py tests/test_round2_gpu.py -m gpu -q -x -p no:cacheprovider \
    -k "gamma_beta or graph_replay or near_zero or signature or model_cases or dkey" > $O/r03_a_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03_a_pytest.log; tail -3 $O/r03_a_pytest.log
python tools/gemv_bench.py > $O/r03_gemv_bench.json 2> $O/r03_gemv_bench.err; cat $O/r03_gemv_bench.json
python tools/gemv_bench.py --flush > $O/r03_gemv_bench_flush.json 2>> $O/r03_gemv_bench.err; cat $O/r03_gemv_bench_flush.json
# one-rank nccl rehearsal: no exchange / staged / unstaged
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-stress --no-kernel-timing > $O/r03_reh_n1.json 2> $O/r03_reh.err
for mode in "" "--unstaged"; do
  DEEPIPR_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
    bench.py --gpus 1 --steps 60 --warmup 10 --no-cpu-baseline --no-stress --no-kernel-timing $mode >> $O/r03_reh_ddp1.jsonl 2>> $O/r03_reh.err
done
DEEPIPR_FORCE_DDP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 \
    bench.py --gpus 1 --steps 60 --warmup 10 --scheme 2 --classes 100 --batch 32 --no-cpu-baseline --no-stress --no-kernel-timing >> $O/r03_reh_ddp1.jsonl 2>> $O/r03_reh.err
python bench.py --steps 60 --warmup 10 --scheme 2 --classes 100 --batch 32 --no-cpu-baseline --no-stress --no-kernel-timing >> $O/r03_reh_n1.json 2>> $O/r03_reh.err
cat $O/r03_reh_n1.json $O/r03_reh_ddp1.jsonl | cut -c1-400
# two ranks sharing the one GPU over gloo, typed without a launcher (functional rehearsal of the self-launcher)
DEEPIPR_SHARE_GPU=1 DEEPIPR_ALLOW_SYNC=0 timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-stress --no-kernel-timing \
    > $O/r03_selflaunch_2rank_gloo.json 2> $O/r03_selflaunch.err; echo "selflaunch rc=$?"; cut -c1-600 $O/r03_selflaunch_2rank_gloo.json; tail -5 $O/r03_selflaunch.err
# vendor convolution determinism
for mode in "" "--perturb" "--prime" "--prime --perturb" "--find"; do
  timeout 400 python tools/conv_determinism.py --reps 200 $mode >> $O/r03_conv_determinism.jsonl 2>> $O/r03_conv_determinism.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_conv_determinism.jsonl'):
    r = json.loads(l); print(r['mode'], r['primed'], r['perturb'], r['nondeterministic'])
PY
cd /tmp && rm -rf /tmp/pin && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pin -o pin -- python $GRAFT_REPO_ROOT/tools/pinned_step.py --private > /tmp/pin.log 2>&1
cp /tmp/pin/pin_kernel_stats.csv $GRAFT_REPO_ROOT/$O/r03_pinned_step_private_kernel_stats.csv 2>/dev/null || (ls -R /tmp/pin | head; tail -5 /tmp/pin.log)
cut -d, -f1-4 $GRAFT_REPO_ROOT/$O/r03_pinned_step_private_kernel_stats.csv | head -50
