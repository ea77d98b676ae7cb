#!/bin/bash
# round 3, pass D: steady-state kernel traces (with idle-gap analysis) of the one-graph, two-graph and four-graph step;
# large-map channel-range passes; full suite
cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29641"
for v in graph1 staged-2-noxchg staged-4; do
  cd /tmp && rm -rf /tmp/tr_$v && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- $RUN $GRAFT_REPO_ROOT/tools/staged_probe.py --only $v --steps 30 > /tmp/tr_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/tr_$v -name "*kernel_trace.csv" | head -1)
  python tools/trace_summary.py "$f" --steps 20 --top 12 --gaps 14 > $O/r03_trace_$v.md 2>&1
  head -3 $O/r03_trace_$v.md; grep -A16 "idle between" $O/r03_trace_$v.md | cut -c1-220
done
python -m pytest tests/test_round3_gpu.py -m gpu -q -x -p no:cacheprovider -k "large_maps or cross_entropy" > $O/r03_d_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03_d_pytest.log; tail -4 $O/r03_d_pytest.log
for cfg in "--image-size 224 --classes 1000" "--arch resnet50 --image-size 224 --classes 1000 --batch 64"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stress $cfg >> $O/r03_bench_large.jsonl 2>> $O/r03_bench_large.err
done
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench_large.jsonl'):
    d = json.loads(l); print(d['config']['workload'][:70], d['ms_per_step'], d['value'], d.get('roofline', {}).get('kernel', '')[:30], d.get('roofline', {}).get('frac'))
    print({k: (v['launches_per_step'], v['us_per_step'], v.get('frac')) for k, v in d.get('kernels', {}).items()})
PY
