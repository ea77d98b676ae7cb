#!/bin/bash
# round 3, pass G: which earlier test makes the graph-vs-eager equality tests fail?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
FIVE="graphed_step_equals_eager_step or trainer_graph_mode_equals_eager_epoch or lr_schedule or survives_an_eager_step"
run() { name=$1; shift; python -m pytest tests -m gpu -q -p no:cacheprovider -k "$*" > $O/r03_g_$name.log 2>&1; echo "$name: $(tail -1 $O/r03_g_$name.log)"; }
run alone "$FIVE"
run after_event "external_event or $FIVE"
run after_entry "entry_points_run_on_the_gpu or $FIVE"
run after_staged "graph_replay_with_eager or $FIVE"
run after_wholenet "whole_net or $FIVE"
python tools/lr_probe.py --reps 6 > $O/r03_lr_probe.json 2>/dev/null; cut -c1-600 $O/r03_lr_probe.json
