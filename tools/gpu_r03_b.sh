#!/bin/bash
# round 3, pass B: LDS-staged batched GEMV sweep, where the staged step's time goes, new parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out
export TMPDIR=/tmp
T=deepipr_amd/csrc/libdeepipr_hip_trace.so
DEEPIPR_LIB=$T python tools/gemv_bench.py > $O/r03_gemv_sweep.json 2> $O/r03_gemv_sweep.err
DEEPIPR_LIB=$T python tools/gemv_bench.py --flush > $O/r03_gemv_sweep_flush.json 2>> $O/r03_gemv_sweep.err
python tools/gemv_bench.py > $O/r03_gemv_bench.json 2>> $O/r03_gemv_sweep.err
python - <<'PY'
import json
for f in ('r03_gemv_sweep.json', 'r03_gemv_sweep_flush.json', 'r03_gemv_bench.json'):
    d = json.load(open('gpurun_out/' + f))
    for k, v in d.items():
        print(f, k, v if not isinstance(v, dict) else (v['us_per_rep'], v['GBps'], v['frac_of_8TBps']))
PY
RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29621"
$RUN tools/staged_probe.py > $O/r03_staged_probe_v1.json 2> $O/r03_staged_probe.err; tail -1 $O/r03_staged_probe_v1.json
$RUN tools/staged_probe.py --private > $O/r03_staged_probe_v2.json 2>> $O/r03_staged_probe.err; tail -1 $O/r03_staged_probe_v2.json
tail -3 $O/r03_staged_probe.err
python -m pytest tests/test_round3_gpu.py tests/test_integration_gpu.py tests/test_round2_gpu.py tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -s \
    -k "round3 or integration or exchange_timeout or graph_replay or model_cases or gamma_beta or near_zero or fullsize or full_size" > $O/r03_b_pytest.log 2>&1
echo "pytest rc=$?" >> $O/r03_b_pytest.log; grep -i "whole-net\|passed\|failed\|rc=" $O/r03_b_pytest.log | tail -8
