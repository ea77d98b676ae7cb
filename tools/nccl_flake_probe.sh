#!/bin/bash
# How often does the RCCL watchdog die in the one-rank graph-replay exchange test body, under which environment?
#   tools/nccl_flake_probe.sh [runs per variant]
n=${1:-5}
export DEEPIPR_FORCE_DDP=1 MIOPEN_USER_DB_PATH=/tmp/miopen_probe MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_BWD_V4R1=0
mkdir -p /tmp/miopen_probe
run() {
  ok=0; bad=0
  for i in $(seq $n); do
    rm -f /tmp/verdict.json
    env "$@" timeout 300 python -m tests.test_parity_gpu $((29600 + RANDOM % 300)) /tmp/verdict.json > /tmp/probe.log 2>&1
    if grep -q '"ok": true' /tmp/verdict.json 2>/dev/null; then ok=$((ok+1)); else bad=$((bad+1)); grep "^variant" /tmp/probe.log | tail -3; grep -i -E "sequence id|last enqueued|last completed|WorkNCCL|SeqNum|capture" /tmp/probe.log | head -8 | cut -c1-260; grep -m1 -o "what():.*" /tmp/probe.log | cut -c1-160; fi
  done
  echo "$* : ok $ok bad $bad"
}
run PROBE=1
run PROBE=2
run PROBE=3
