#!/bin/bash
# pass J: where do the staged step's ~300 us go?  un-profiled variant times in one process, then per-kernel traces
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29661"
$RUN tools/staged_probe.py > $O/r03_j_probe_v1.json 2> $O/r03_j_probe.err; tail -1 $O/r03_j_probe_v1.json
for v in graph1 staged-1 staged-3-noxchg staged-3; do
  cd /tmp && rm -rf /tmp/tr_$v && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$v -o t -- $RUN $GRAFT_REPO_ROOT/tools/staged_probe.py --only $v --steps 30 > /tmp/tr_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/tr_$v -name "*kernel_trace.csv" | head -1)
  python tools/trace_summary.py "$f" --steps 20 --top 10 --gaps 10 --out-csv $O/r03_j_kernels_$v.csv > $O/r03_j_trace_$v.md 2>&1
  head -3 $O/r03_j_trace_$v.md; grep -A11 "idle between" $O/r03_j_trace_$v.md | cut -c1-200
done
python - <<'PY'
import csv
def load(v):
    return {r['kernel']: (float(r['calls_per_step']), float(r['us_per_step'])) for r in csv.DictReader(open('gpurun_out/r03_j_kernels_%s.csv' % v))}
a = load('graph1')
for v in ('staged-1', 'staged-3-noxchg', 'staged-3'):
    b = load(v)
    rows = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k, a.get(k, (0, 0)), b.get(k, (0, 0))) for k in set(a) | set(b)), reverse=True)
    print(v, 'total busy delta %.1f us/step' % sum(r[0] for r in rows))
    for r in rows[:6] + rows[-3:]:
        print('   %+8.1f us  %s  %s -> %s' % (r[0], r[1][:70], r[2], r[3]))
PY
