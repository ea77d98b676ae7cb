#!/bin/bash
# round 3, pass F: the one-graph staged step (external events) -- semantics test, parity, one-rank RCCL rehearsal both
# policies; full suite run 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp ROUND_TAG=r03
O=gpurun_out
python -m pytest tests/test_round3_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k "external_event or graph_replay_with_eager" > $O/r03_f_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03_f_pytest.log; tail -4 $O/r03_f_pytest.log
bash tools/gpu_ddp1.sh
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29651"
COMMON="--gpus 1 --steps 60 --warmup 15 --no-cpu-baseline --no-stress --no-kernel-timing"
for cfg in "" "--scheme 2 --classes 100 --batch 32"; do
  DEEPIPR_OVERLAP_SYNC=0 DEEPIPR_FORCE_DDP=1 $RUN bench.py $COMMON $cfg 2>&1 | grep -E '"metric"' | sed 's/^{/{"rehearsal": "exchange forced on, exclusive policy (graph split before the split-channel stage)", /' >> $O/r03_ddp_rehearsal.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_ddp_rehearsal.jsonl'):
    d = json.loads(l)
    print('%-90s %-30s %8.1f img/s %7.3f ms exposed %s' % (d['rehearsal'], d['config']['workload'][:30], d['value'], d['ms_per_step'], d.get('exchange_us_exposed')))
PY
DEEPIPR_SHARE_GPU=1 DEEPIPR_ALLOW_SYNC=0 timeout 600 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 --no-cpu-baseline --no-stress --no-kernel-timing \
    > $O/r03_selflaunch_2rank_gloo.json 2> $O/r03_selflaunch.err; echo "selflaunch rc=$?"; cut -c1-300 $O/r03_selflaunch_2rank_gloo.json
rm -f $O/session_end_determinism.json
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r03_pytest_gpu_2.log 2>&1; echo "pytest rc=$?" >> $O/r03_pytest_gpu_2.log; tail -4 $O/r03_pytest_gpu_2.log
cp $O/session_end_determinism.json $O/r03_session_end_run2.json
