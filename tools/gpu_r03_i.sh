#!/bin/bash
# pass I: with the box's default user find-db populated by find-mode processes, do the tests (own, empty database) pass?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
python bench.py --scheme 2 --classes 100 --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-stress --no-kernel-timing > /dev/null 2>&1
python bench.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-stress --no-kernel-timing > /dev/null 2>&1
python -m pytest tests -m gpu -q -p no:cacheprovider -k "graphed_step_equals_eager_step or trainer_graph_mode_equals_eager_epoch or lr_schedule or survives_an_eager_step" > $O/r03_i_five.log 2>&1; tail -1 $O/r03_i_five.log
MIOPEN_USER_DB_PATH=$HOME/.config/miopen python -m pytest tests -m gpu -q -p no:cacheprovider -k "graphed_step_equals_eager_step or trainer_graph_mode_equals_eager_epoch or lr_schedule or survives_an_eager_step" > $O/r03_i_five_shared_db.log 2>&1; tail -1 $O/r03_i_five_shared_db.log
