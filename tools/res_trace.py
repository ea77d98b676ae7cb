#!/usr/bin/env python
"""Where the time goes inside the register-resident single-pass kernels: per-workgroup wall-clock stamps
(deepipr_debug_trace, 100 MHz) of the phases of k_bn_res_fwd / k_bn_res_bwd, next to the kernel duration from the
per-dispatch HIP events.  Needs the measurement build of the library (the production build compiles the stamps out):

    make -C deepipr_amd/csrc trace && DEEPIPR_LIB=deepipr_amd/csrc/libdeepipr_hip_trace.so python tools/res_trace.py [N]

(In that build k_bn_res_bwd<1024, 8> spills 28 B per lane -- its phases are slightly stretched; forward is exact.)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import _lib                                   # noqa: E402
from deepipr_amd.passport_ops import kernels as K              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(N, 64, 32, 32), (N, 128, 16, 16), (N, 256, 8, 8), (N, 512, 4, 4), (4 * N, 512, 8, 8)]
if os.environ.get('DEEPIPR_TRACE_SHAPES'):       # "256,256,56,56;256,64,56,56": maps beyond the register file -- the stamps of a ranges
    SHAPES = [tuple(int(v) for v in sh.split(',')) for sh in os.environ['DEEPIPR_TRACE_SHAPES'].split(';')]    # launch are its LAST range's
dev = torch.device('cuda:0')
TICK_US = 0.01                                                  # wall_clock64(): 100 MHz


def phases(buf, nblk):
    t = buf[:nblk * 8].view(nblk, 8).cpu().numpy().astype(np.int64)
    t = t[t[:, 0] > 0]
    if t[:, 3].max() == 0:                       # backward has no separate table phase: stamp 3 := stamp 2
        t[:, 3] = t[:, 2]
    t0 = t[:, 0].min()
    rel = (t[:, :5] - t0) * TICK_US
    d = np.diff(rel, axis=1)
    return {'workgroups': len(t), 'start_skew': rel[:, 0].max(),
            'load+sums': (d[:, 0].mean(), d[:, 0].max()), 'exchange': (d[:, 1].mean(), d[:, 1].max()),
            'table': (d[:, 2].mean(), d[:, 2].max()), 'stores_issued': (d[:, 3].mean(), d[:, 3].max()),
            'span': rel[:, 4].max()}


def run(shape, reps=10):
    n, c, h, w = shape
    x = torch.randn(shape, device=dev)
    dy = torch.randn(shape, device=dev)
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    buf = torch.zeros(8 * 1100, dtype=torch.int64, device=dev)
    out = None
    for _ in range(3):
        out = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
        K.passport_bn_bwd(dy, x, out[1], None, None, 0.0, None, None, None, None, True, True)
    torch.cuda.synchronize()
    res = {}
    for which in ('fwd', 'bwd'):
        acc = []
        _lib.profile_enable(True)
        for _ in range(reps):
            buf.zero_()
            torch.cuda.synchronize()
            _lib.lib().deepipr_debug_trace(ctypes.c_void_p(buf.data_ptr()))
            if which == 'fwd':
                out = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
            else:
                K.passport_bn_bwd(dy, x, out[1], None, None, 0.0, None, None, None, None, True, True)
            torch.cuda.synchronize()
            _lib.lib().deepipr_debug_trace(None)
            acc.append(phases(buf, 1100))
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        kern = 1000.0 * prof['bn_res_' + which][0] / max(1, prof['bn_res_' + which][1])
        med = acc[len(acc) // 2]
        res[which] = (kern, med)
    return res


if 'trace' not in os.path.basename(_lib.LIB_PATH):
    raise SystemExit('res_trace.py needs the measurement build: see the docstring')
for shape in SHAPES:
    r = run(shape)
    mb = 4 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
    for which in ('fwd', 'bwd'):
        kern, p = r[which]
        print('%-20s %5.1f MB %s kernel %6.2f us | %4d WGs start skew %5.2f | load+sums %5.2f (max %5.2f) | exchange %5.2f '
              '(max %5.2f) | table %4.2f | stores issued %5.2f (max %5.2f) | entry->last store issued %6.2f us'
              % (shape, mb, which, kern, p['workgroups'], p['start_skew'], p['load+sums'][0], p['load+sums'][1],
                 p['exchange'][0], p['exchange'][1], p['table'][0], p['stores_issued'][0], p['stores_issued'][1],
                 p['span']))
