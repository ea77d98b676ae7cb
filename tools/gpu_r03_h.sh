#!/bin/bash
# round 3, pass H: does a populated MIOpen user find-db (left by earlier find-mode processes on the box) make the pinned
# bs-32 convolutions nondeterministic?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
ls -la ~/.config/miopen 2>/dev/null | head -5
python tools/conv_determinism.py --batches 32 --reps 300 > $O/r03_h_conv_fresh.json 2>/dev/null; python -c "import json;d=json.load(open('$O/r03_h_conv_fresh.json'));print('fresh db:', d['nondeterministic'])"
python bench.py --scheme 2 --classes 100 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-stress --no-kernel-timing > /dev/null 2>&1
python bench.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-stress --no-kernel-timing > /dev/null 2>&1
ls -la ~/.config/miopen 2>/dev/null | head -8; find ~/.config/miopen -type f | head; du -sh ~/.config/miopen 2>/dev/null
python tools/conv_determinism.py --batches 32 --reps 300 > $O/r03_h_conv_after_find.json 2>/dev/null; python -c "import json;d=json.load(open('$O/r03_h_conv_after_find.json'));print('after find-mode processes:', d['nondeterministic']); print({k:v for k,v in d['configs'].items() if any(v.values())})"
python tools/lr_probe.py --reps 6 > $O/r03_h_lr_probe.json 2>/dev/null; cut -c1-1500 $O/r03_h_lr_probe.json
python -m pytest tests -m gpu -q -p no:cacheprovider -k "graphed_step_equals_eager_step or trainer_graph_mode_equals_eager_epoch or lr_schedule or survives_an_eager_step" > $O/r03_h_five.log 2>&1; grep -n "worst\|passed\|failed" $O/r03_h_five.log | cut -c1-600
