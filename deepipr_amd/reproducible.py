"""Bit-reproducible train steps: what has to be pinned OUTSIDE this library.

This library's kernels are bit-reproducible by construction (fixed-order reductions, no float atomics).  A whole
train step is reproducible only as far as the vendor convolution is (profiles/r03_determinism.md):
  * find mode (cudnn.benchmark = True) measures solvers at run time and may pick different ones from run to run;
  * immediate mode (cudnn.benchmark = False, cudnn.deterministic = True) takes the first solution MIOpen offers -- which,
    once the user find-database holds a record for the problem, is that record's fastest solver, and torch's
    deterministic flag does not filter it.  For some backward-data problems of ResNet18 that is composable-kernel's
    grouped backward-data solver: split-K with atomic accumulation, different in the last bit from run to run.
`pin()` sets what makes a step reproducible on this stack: immediate mode, and the atomic solver switched off by
MIOpen's own per-solver switch.  Call it BEFORE the first convolution of the process (MIOpen reads the variable when it
first enumerates solvers).  The reference has no counterpart (train_v1.py:8 only sets cudnn.benchmark = True).
"""
import os

# MIOpen's switch for ConvHipImplicitGemmGroupBwdXdlops (grouped backward-data through composable kernel).
ENV = {'MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_GROUP_BWD_XDLOPS': '0'}


def pin(empty_find_db=False):
    """Make the vendor convolutions of this process run-to-run reproducible.  empty_find_db=True also points the user
    find-database at a fresh directory, so that what ran on the machine before cannot change which solvers are used."""
    import torch
    for k, v in ENV.items():
        os.environ[k] = v
    if empty_find_db and 'MIOPEN_USER_DB_PATH' not in os.environ:
        import tempfile
        os.environ['MIOPEN_USER_DB_PATH'] = tempfile.mkdtemp(prefix='deepipr_miopen_udb_')
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True
