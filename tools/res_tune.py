#!/usr/bin/env python
"""Sweep the planning knobs of the register-resident single-pass kernels (deepipr_debug_tune) per shape.
GPU box only:  python tools/res_tune.py [N]        prints fwd / bwd kernel microseconds (per-dispatch HIP events)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import _lib                                   # noqa: E402
from deepipr_amd.passport_ops import kernels as K              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(N, 64, 32, 32), (N, 128, 16, 16), (N, 256, 8, 8), (N, 512, 4, 4), (4 * N, 512, 8, 8)]
dev = torch.device('cuda:0')
CONFIGS = [('default', {}), ('xcd_map on', {'xcd_map': 1}), ('split below half only', {'split_full': 0}),
           ('always 1024 threads', {'small_t': 0})]
KNOBS = {'split_full': 1, 'xcd_map': 0, 'small_t': 1}


def run(shape, tail, reps=30):
    n, c, h, w = shape
    x = torch.randn(shape, device=dev)
    dy = torch.randn(shape, device=dev)
    dy2 = torch.randn(shape, device=dev) if tail else None
    res = torch.randn(shape, device=dev) if tail else None
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)

    def once():
        out = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True, residual=res)
        K.passport_bn_bwd(dy, x, out[1], None, None, 0.0, None, None, None, None, True, True, dy2=dy2,
                          tail_out=out[0] if tail else None)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    assert prof['bn_res_fwd'][1] == reps and prof['bn_res_bwd'][1] == reps, 'shape left the single-pass form'
    return 1000.0 * prof['bn_res_fwd'][0] / reps, 1000.0 * prof['bn_res_bwd'][0] / reps


for tail in (False, True):
    for shape in SHAPES:
        mb = 4 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
        fb, bb = (12.0, 24.0) if tail else (8.0, 12.0)
        print('%s %-22s %6.1f MB' % ('tail ' if tail else 'plain', shape, mb))
        for name, knobs in CONFIGS:
            for k, default in KNOBS.items():
                _lib.debug_tune(k, knobs.get(k, default))
            try:
                f, b = run(shape, tail)
                print('    %-22s fwd %6.2f us (%5.2f TB/s)  bwd %6.2f us (%5.2f TB/s)' %
                      (name, f, fb / 4 * mb / f, b, bb / 4 * mb / b))
            except (AssertionError, RuntimeError) as e:
                print('    %-22s -- %s' % (name, str(e)[:80]))
for k, default in KNOBS.items():
    _lib.debug_tune(k, default)
print('sync timeouts:', K.sync_timeouts())
