#!/bin/bash
# Round-6 evidence in one gpurun call (results in gpurun_out/r06/; copy what is to be judged into profiles/):
#   1. smoke()                                                    -> smoke.log
#   2. bench.py exactly as the driver runs it (no flags)           -> bench_n1.json  (wall time of the whole run beside it)
#   3. rocprofv3 --kernel-trace --stats of the config-R step       -> steady_state.md, kernel_stats_top.csv, bench_under_rocprof.json
#   4. the same for ResNet50 at ImageNet geometry, batch 256       -> steady_state_r50.md
#   5. the standalone ResNet50 bench line with per-kernel timing   -> bench_r50_bs256.json
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
t0=$(date +%s)
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench_n1.json
echo "bench.py default run: $(( $(date +%s) - t0 )) s wall" | tee $O/bench_n1_wall.txt
cut -c1-400 $O/bench_n1.json
cd /tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r06 -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-stress --no-configs > /tmp/prof.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof/r06_kernel_trace.csv --steps 40 --top 60 > $O/steady_state.md
head -60 /tmp/prof/r06_kernel_stats.csv | cut -c1-400 > $O/kernel_stats_top.csv
grep '"metric"' /tmp/prof.log | cut -c1-6000 > $O/bench_under_rocprof.json
head -3 $O/steady_state.md
rm -rf /tmp/prof50 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof50 -o r50 -- python $GRAFT_REPO_ROOT/bench.py --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 40 --warmup 10 --no-cpu-baseline --no-stress --no-configs --no-kernel-timing > /tmp/prof50.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py /tmp/prof50/r50_kernel_trace.csv --steps 40 --top 80 > $O/steady_state_r50.md
head -60 /tmp/prof50/r50_kernel_stats.csv | cut -c1-400 > $O/kernel_stats_top_r50.csv
head -3 $O/steady_state_r50.md
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --arch resnet50 --image-size 224 --classes 1000 --batch 256 --no-miopen-find --steps 20 --warmup 5 --no-cpu-baseline --no-stress --no-configs 2>/dev/null | tail -1 > $O/bench_r50_bs256.json
cut -c1-300 $O/bench_r50_bs256.json
