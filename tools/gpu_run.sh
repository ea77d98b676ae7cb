#!/bin/bash
# ONE parameterised entry for the GPU box (replaces round 3's twenty one-shot tools/gpu_r03_[a-t].sh, kept in git history).
#   tools/gpu_run.sh suite [tag] [pytest args]       full `-m gpu` session -> gpurun_out/pytest_gpu_<tag>.log
#   tools/gpu_run.sh tests <-k expression>           a subset, first failure stops
#   tools/gpu_run.sh ab VAR=value [reps]             A/B of one environment switch on configs R / P shard / AlexNet
#   tools/gpu_run.sh steady <tag> [bench args]       rocprofv3 kernel trace of the replayed step -> gpurun_out/<tag>.md
#   tools/gpu_run.sh profiles <round tag>            the round's measurement set (steady states, ImageNet-shape lines)
#   tools/gpu_run.sh bench <tag>                     bench lines R / V3 / P shard with their roofline objects
#   tools/gpu_run.sh conv                            conv kernels (wgrad / fwd / dgrad) against the vendor library, per shape
#   tools/gpu_run.sh nrank                           the N > 1 code path of bench.py rehearsed on one GPU -> gpurun_out/<round>_nrank_rehearsal.jsonl
#   tools/gpu_run.sh micro                           tools/micro/mfma_rate.hip: issue interval of the bf16 MFMA alone and beside vector instructions
#   tools/gpu_run.sh whatif                          Winograd kernels with parts of the main loop left out (build first: tools/wino_whatif.sh build) -> gpurun_out/wino_whatif.jsonl
#   tools/gpu_run.sh forms [batch]                   every planner form of the Winograd forward / backward-data kernels forced over the ResNet18 shapes -> gpurun_out/wino_forms.jsonl
#   tools/gpu_run.sh prebench [batch]                in-kernel weight transform against the pre-transformed form, per layer shape
#   tools/gpu_run.sh branches                        tools/micro/graph_branches.py: do independent branches of a replayed hipGraph run concurrently
# Run through gpurun from the repo root:  gpurun --timeout 1800 -- 'tools/gpu_run.sh suite r04_final'
cmd=$1; shift
case "$cmd" in
  suite)    exec tools/gpu_suite.sh "$@" ;;
  tests)    exec timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$1" ;;
  ab)       exec tools/gpu_ab.sh "$@" ;;
  steady)   exec tools/gpu_steady.sh "$@" ;;
  profiles) exec tools/gpu_round_profiles.sh "$@" ;;
  bench)    exec tools/gpu_bench_check.sh "$@" ;;
  conv)     python tools/wgrad_bench.py --json gpurun_out/wgrad_bench.json | grep "^{"; exec python tools/conv_bench.py --json gpurun_out/conv_bench.json ;;
  nrank)    exec tools/gpu_nrank_rehearsal.sh "$@" ;;
  whatif)   exec tools/wino_whatif.sh run "$@" ;;
  forms)    WHATIF_BATCH=${1:-128} exec tools/wino_forms.sh ;;
  prebench) exec python tools/wino_pre_bench.py --batch ${1:-128} ;;
  branches) exec python tools/micro/graph_branches.py ;;
  micro)    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/mfma_rate.out tools/micro/mfma_rate.hip && exec /tmp/mfma_rate.out ;;
  *)        sed -n 2,17p "$0"; exit 2 ;;
esac
