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
    # the form the step runs: weights pre-transformed (k_wino_weights), DEEPIPR_TRACE_RAW=1 for the in-kernel transform
    pre = None if os.environ.get('DEEPIPR_TRACE_RAW') == '1' else K.wino_transform([w])[0]
    fn = (lambda: K.conv_fwd(x, w, 1, 1, pre)) if direction == 0 else (lambda: K.conv_dgrad(x, w, x.shape, 1, 1, pre))
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
    # effective shader clock while the kernel runs: shader cycles over wall time (100 MHz counter), entry -> stores issued, per workgroup
    dcyc = (t[:, 4, 1] - t[:, 0, 1]).astype(np.float64)
    dwall = (t[:, 4, 0] - t[:, 0, 0]).astype(np.float64) * 0.01
    line += ' | shader clock %4.0f MHz (min %4.0f)' % ((dcyc / dwall).mean(), (dcyc / dwall).min())
    chunks = c // 8
    line += ' | %5.0f cyc/chunk' % (cyc[:, 1].mean() / max(1, chunks))
    line += ' | wall: loop starts %5.1f..%5.1f, ends %5.1f..%5.1f' % (wall[:, 1].min(), wall[:, 1].max(), wall[:, 2].min(), wall[:, 2].max())
    print(line, flush=True)
    if os.environ.get('DEEPIPR_TRACE_PLACEMENT') == '1':
        # which workgroups share a CU: (XCC, HW_ID bits 8-14) per workgroup index, the first 2 x CUs of them (the first round)
        full = buf.view(-1, 32, 2).cpu().numpy().astype(np.int64)[:len(t)]
        where = full[:, 31, 1] * 128 + full[:, 31, 0]
        first = {}
        for b in range(min(len(t), 512)):
            first.setdefault(int(where[b]), []).append(b)
        pairs = [v for v in first.values() if len(v) >= 2]
        d = sorted({v[1] - v[0] for v in pairs})
        print('      placement: %d distinct CUs among the first 512 workgroups, %d hold two or more; index distance of a CU\'s first two tenants: %s; examples %s'
              % (len(first), len(pairs), d[:12], pairs[:4]), flush=True)
        print('      XCC of workgroups 0..15:', full[:16, 31, 1].tolist(), ' HW_ID[14:8] of 0..15:', full[:16, 31, 0].tolist(), flush=True)
    if not t[:, 5, 1].any():                                   # the pre-transformed form carries no per-step stamps
        return
    steps = min(9, chunks)
    st = t[:, 5:5 + 3 * steps, 1].reshape(len(t), steps, 3)
    body, bar = (st[:, :, 1] - st[:, :, 0]), (st[:, :, 2] - st[:, :, 1])
    gap = st[:, 1:, 0] - st[:, :-1, 2]
    print('      per step (cycles, mean over workgroups): body', np.round(body.mean(0)).astype(int).tolist(), 'barrier wait', np.round(bar.mean(0)).astype(int).tolist(),
          'gap', np.round(gap.mean(0)).astype(int).tolist(), flush=True)


for n, c, hw in SHAPES:
    for d in (0, 1):
        run(n, c, hw, d)
