#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace) of one tool:  tools/gpu_kprof.sh <pattern> <python tool and args...>
pat=$1; shift
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/kp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kp -o k -- python $GRAFT_REPO_ROOT/"$@" > /tmp/kp.log 2>&1
grep "^{" /tmp/kp.log | cut -c1-160 | head -20
python - "$pat" <<'PY'
import collections, csv, sys
st = collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/kp/k_kernel_trace.csv')):
    if sys.argv[1] in r['Kernel_Name']:
        st[r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:70]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(st.items()):
    print('%-70s calls %4d avg %7.1f us min %7.1f' % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
