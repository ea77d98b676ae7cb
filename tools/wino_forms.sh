#!/bin/bash
# The planner's forms (m-tile height, k-groups, K splits) of the Winograd forward / backward-data kernels forced one by one
# (DEEPIPR_WINO_FORM, measurement knob of plan_conv_wino) over the four ResNet18 shapes at batch 128 -> gpurun_out/wino_forms.jsonl
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; : > gpurun_out/wino_forms.jsonl
for f in ${FORMS:-default 1,2,1 2,2,1 2,2,2 1,2,2 2,1,2 1,1,2 2,1,1 1,1,1 2,2,4 1,2,4}; do
  if [ $f = default ]; then unset DEEPIPR_WINO_FORM; else export DEEPIPR_WINO_FORM=$f; fi
  WHATIF_MASK=0 timeout 300 python tools/wino_whatif.py | grep '^{' | sed "s/^{/{\"form\": \"$f\", /" | tee -a gpurun_out/wino_forms.jsonl
done
