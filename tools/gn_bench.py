#!/usr/bin/env python
"""Per-shape timing of the GroupNorm / InstanceNorm-fused kernels (in-situ per-dispatch events) next to the
BatchNorm single-pass kernels on the same tensors.  GPU box only:  python tools/gn_bench.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import _lib                                   # noqa: E402
from deepipr_amd.passport_ops import kernels as K              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(N, 64, 32, 32), (N, 128, 16, 16), (N, 256, 8, 8), (N, 512, 4, 4)]
dev = torch.device('cuda:0')


def run(shape, groups, reps=40):
    n, c, h, w = shape
    x = torch.randn(shape, device=dev)
    dy = torch.randn(shape, device=dev)
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)

    def once():
        if groups:
            out = K.passport_gn_fwd(x, None, None, g, b, None, 0.0, True, groups, 1e-5)
            K.passport_gn_bwd(dy, x, out[1], g, b, None, None, 0.0, None, None, None, None, True, groups)
        else:
            out = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
            K.passport_bn_bwd(dy, x, out[1], None, None, 0.0, None, None, None, None, True, True)
    for _ in range(5):
        once()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    us = lambda k: 1000.0 * prof[k][0] / reps
    if groups:
        return us('gn_fwd'), us('gn_bwd'), us('reduce_partials')
    return us('bn_res_fwd'), us('bn_res_bwd'), 0.0


for shape in SHAPES:
    c = shape[1]
    mb = 4 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
    line = '%-20s %5.1f MB' % (shape, mb)
    for name, groups in (('bn', 0), ('gn', c // 16), ('in', c)):
        f, b, r = run(shape, groups)
        line += ' | %s fwd %6.2f bwd %6.2f (+%.1f finish) us' % (name, f, b, r)
    print(line)
