#!/bin/bash
# wgrad kernel bench under rocprofv3 (per-kernel durations), then the plain bench.   tools/gpu_wgrad.sh [tag] [bench args]
tag=${1:-x}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/wg_prof && rocprofv3 --kernel-trace --output-format csv -d /tmp/wg_prof -o wg -- python $GRAFT_REPO_ROOT/tools/wgrad_bench.py --reps 20 "$@" > /tmp/wg_prof.log 2>&1
python - /tmp/wg_prof/wg_kernel_trace.csv <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/wgrad_stats_${tag}.txt
import collections, csv, sys
st = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if 'wgrad' in n or 'igemm' in n or 'transpose' in n or 'SubTensor' in n or 'Winograd' in n.lower() or 'miopen' in n.lower():
        st[(n[:70], r.get('Grid_Size', r.get('Grid_Size_X', '')))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for (n, g), v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
    v = v[len(v) // 4:]                      # drop warm-up / find-mode calls
    print('%-70s grid %-9s calls %4d avg %7.1f us min %7.1f' % (n, g, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
PY
cd $GRAFT_REPO_ROOT
python tools/wgrad_bench.py "$@" --json gpurun_out/wgrad_bench_${tag}.json 2>&1 | grep "^{"
