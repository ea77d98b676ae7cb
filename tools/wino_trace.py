#!/usr/bin/env python
"""Where the time goes inside k_conv_wino: per-workgroup stamps (wall clock 100 MHz + shader clock) at entry / first chunk
staged / MFMA loop done / Z parked / stores issued, next to the per-dispatch duration.  Measurement build only:

    make -C deepipr_amd/csrc trace && DEEPIPR_LIB=deepipr_amd/csrc/libdeepipr_hip_trace.so python tools/wino_trace.py [N]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import _lib                                   # noqa: E402
from deepipr_amd.passport_ops import kernels as K              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(N, 64, 32), (N, 128, 16), (N, 256, 8), (N, 512, 4)]
dev = torch.device('cuda:0')


def run(n, c, hw, direction):
    x = torch.randn(n, c, hw, hw, device=dev)
    w = torch.randn(c, c, 3, 3, device=dev) * 0.05
    buf = torch.zeros(64 * 8192, dtype=torch.int64, device=dev)
    fn = (lambda: K.conv_fwd(x, w, 1, 1)) if direction == 0 else (lambda: K.conv_dgrad(x, w, x.shape, 1, 1))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    _lib.lib().deepipr_debug_wino_trace(ctypes.c_void_p(buf.data_ptr()))
    _lib.profile_enable(True)
    fn()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    _lib.lib().deepipr_debug_wino_trace(None)
    prof = _lib.profile_read()
    us = 1000.0 * sum(prof[k][0] for k in ('conv_wino_fwd', 'conv_wino_dgrad'))
    t = buf.view(-1, 32, 2).cpu().numpy().astype(np.int64)
    t = t[t[:, 0, 0] > 0]
    wall = (t[:, :, 0] - t[:, 0, 0].min()) * 0.01              # us since the first workgroup's entry
    cyc = np.diff(t[:, :5, 1], axis=1)                           # shader cycles per phase
    names = ['prologue', 'mfma_loop', 'park', 'store']
    line = '%4d x %3d @%2d %s: %6.1f us, %4d wgs, start skew %5.1f us, last end %5.1f us |' % (
        n, c, hw, 'fwd' if direction == 0 else 'dgr', us, len(t), wall[:, 0].max(), wall[:, 4].max())
    for i, nm in enumerate(names):
        line += ' %s %6.0f cyc (max %6.0f)' % (nm, cyc[:, i].mean(), cyc[:, i].max())
    chunks = c // 8
    line += ' | %5.0f cyc/chunk' % (cyc[:, 1].mean() / max(1, chunks))
    line += ' | wall: loop starts %5.1f..%5.1f, ends %5.1f..%5.1f' % (wall[:, 1].min(), wall[:, 1].max(), wall[:, 2].min(), wall[:, 2].max())
    print(line, flush=True)
    steps = min(9, chunks)
    st = t[:, 5:5 + 3 * steps, 1].reshape(len(t), steps, 3)
    body, bar = (st[:, :, 1] - st[:, :, 0]), (st[:, :, 2] - st[:, :, 1])
    gap = st[:, 1:, 0] - st[:, :-1, 2]
    print('      per step (cycles, mean over workgroups): body', np.round(body.mean(0)).astype(int).tolist(), 'barrier wait', np.round(bar.mean(0)).astype(int).tolist(),
          'gap', np.round(gap.mean(0)).astype(int).tolist(), flush=True)


for n, c, hw in SHAPES:
    for d in (0, 1):
        run(n, c, hw, d)
