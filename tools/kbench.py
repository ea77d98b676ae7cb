#!/usr/bin/env python
"""Micro-benchmark of the C-ABI kernels at the workload's and at stress shapes.

Each kernel is launched `reps` times back to back between two HIP events (so the figure includes the
~1.5 us kernel boundary); run the script under `rocprofv3 --kernel-trace --stats` for pure kernel
durations.  Prints one line per (kernel, shape): us per launch and algorithmic GB/s.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepipr_amd.passport_ops import kernels as K  # noqa: E402

SHAPES = {
    'R   [128,512,4,4]': (128, 512, 4, 4),
    'A   [64,384,8,8]': (64, 384, 8, 8),
    'P   [32,512,4,4]': (32, 512, 4, 4),
    'I   [256,512,7,7]': (256, 512, 7, 7),
    'L3  [128,256,8,8]': (128, 256, 8, 8),
    'L2  [128,128,16,16]': (128, 128, 16, 16),
    'S1  [128,64,32,32]': (128, 64, 32, 32),
    'S2  [256,256,16,16]': (256, 256, 16, 16),
    'S3  [512,512,8,8]': (512, 512, 8, 8),
}


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1000.0 / reps


def main():
    dev = 'cuda:0'
    only = sys.argv[1:] or None
    res = []
    for name, shp in SHAPES.items():
        if only and not any(o in name for o in only):
            continue
        n, c, h, w = shp
        x = torch.randn(shp, device=dev)
        dy = torch.randn(shp, device=dev)
        g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
        el = x.numel()
        t = timed(lambda: K.affine_relu_fwd(x, g, b, True))
        res.append(('affine_fwd', name, t, 8 * el / t / 1e3))
        t = timed(lambda: K.affine_relu_bwd(dy, x, g, b, True))
        res.append(('affine_bwd(+finish)', name, t, 12 * el / t / 1e3))
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)

        def fused():
            out = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
            K.passport_bn_bwd(dy, x, out[1], None, None, 0.0, None, None, None, None, True, True)
        t = timed(fused)
        res.append(('bn fused fwd+bwd (32B)', name, t, 32 * el / t / 1e3))
        cpy = torch.empty_like(x)
        t = timed(lambda: cpy.copy_(x))
        res.append(('torch copy_ (8B/elt)', name, t, 8 * el / t / 1e3))
        t = timed(lambda: torch.relu(x))
        res.append(('torch relu (8B/elt)', name, t, 8 * el / t / 1e3))
    for co, kk in ((512, 4608), (512, 2304), (512, 256), (384, 1728)):
        if only and 'gb' not in only:
            continue
        wt = torch.randn(co, kk, device=dev)
        m = torch.rand(2, kk, device=dev, dtype=torch.float64)
        dg, db = torch.randn(co, device=dev), torch.randn(co, device=dev)
        t = timed(lambda: K.gamma_beta_fwd(wt, m))
        res.append(('gamma_beta_fwd', 'W[%d,%d]' % (co, kk), t, 4 * co * kk / t / 1e3))
        t = timed(lambda: K.gamma_beta_bwd(dg, db, m, (co, kk)))
        res.append(('gamma_beta_bwd', 'W[%d,%d]' % (co, kk), t, 4 * co * kk / t / 1e3))
    for r in res:
        print('%-22s %-22s %9.2f us  %8.1f GB/s' % r)


if __name__ == '__main__':
    main()
