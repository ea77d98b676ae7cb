#!/bin/bash
# What does each part of the Winograd kernels' main loop cost?  Measurement builds with parts LEFT OUT (WN_WHATIF bit mask,
# deepipr_conv_wino.inc; the results are wrong on purpose, only the time is read).
#   tools/wino_whatif.sh build [masks...]     (build container) -> deepipr_amd/csrc/obj/whatif/libwi_<mask>.so
#   tools/wino_whatif.sh run [masks...]       (GPU box)         -> gpurun_out/wino_whatif.jsonl
cd "$(dirname "$0")/.." || exit 1
mode=$1; shift
masks=${*:-"0 1 2 4 8 16 32 64 14 30 62 126"}
dir=deepipr_amd/csrc/obj/whatif
case "$mode" in
  build)
    make -C deepipr_amd/csrc >/dev/null || exit 1
    mkdir -p $dir
    for m in $masks; do
      ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -w -DWN_WHATIF=$m \
          -c -o $dir/wino_$m.o deepipr_amd/csrc/deepipr_wino.hip &&
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $dir/libwi_$m.so deepipr_amd/csrc/obj/main.o $dir/wino_$m.o && rm $dir/wino_$m.o ) &
    done
    wait; ls -la $dir ;;
  run)
    mkdir -p gpurun_out; : > gpurun_out/wino_whatif.jsonl
    for m in $masks; do
      DEEPIPR_LIB=$PWD/$dir/libwi_$m.so WHATIF_MASK=$m timeout 300 python tools/wino_whatif.py | grep '^{' | tee -a gpurun_out/wino_whatif.jsonl
    done ;;
esac
