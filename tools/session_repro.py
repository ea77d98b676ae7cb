"""Reproduce session-dependent failures: run the non-pinned GPU tests in THIS process (what a full session has behind
it when the pinned tests start), then the graph-vs-eager trajectory tests several times over, in the same process.
    python tools/session_repro.py [--reps 5]"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
reps = int(sys.argv[sys.argv.index('--reps') + 1]) if '--reps' in sys.argv else 5
GRAPH = ('graphed_step_equals_eager_step or trainer_graph_mode_equals_eager_epoch or lr_schedule or survives_an_eager_step '
         'or reads_gradients_in_place or graph_replay_with_eager')
rc = pytest.main(['tests', '-m', 'gpu and not miopen_pinned', '-q', '-p', 'no:cacheprovider'])
print('prelude rc', rc, flush=True)
for i in range(reps):
    rc = pytest.main(['tests', '-m', 'gpu', '-q', '-p', 'no:cacheprovider', '-k', GRAPH])
    print('graph tests, repetition', i, 'rc', rc, flush=True)
