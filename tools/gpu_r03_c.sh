#!/bin/bash
# round 3, pass C: bisect the lr-schedule replay mismatch; kernel traces of the one-graph and the two-graph step
cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
for env in "" "DEEPIPR_NO_GEMV_BATCH=1"; do
  echo "== lr test, $env" >> $O/r03_c_lr.log
  env $env python -m pytest tests/test_round2_gpu.py -m gpu -q -x -p no:cacheprovider -k "lr_schedule or reads_gradients_in_place" >> $O/r03_c_lr.log 2>&1
  tail -2 $O/r03_c_lr.log
done
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29631"
for v in graph1 staged-2-noxchg staged-4; do
  cd /tmp && rm -rf /tmp/tr_$v && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -o t -- $RUN $GRAFT_REPO_ROOT/tools/staged_probe.py --only $v --steps 40 > /tmp/tr_$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find /tmp/tr_$v -name "*kernel_stats.csv" | head -1)
  cp "$f" $O/r03_trace_${v}_kernel_stats.csv 2>/dev/null
  tail -1 /tmp/tr_$v.log | cut -c1-300
  python - "$O/r03_trace_${v}_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print(sys.argv[1], 'kernels', len(rows), 'calls', calls, 'total_ms', round(tot / 1e6, 2))
PY
done
