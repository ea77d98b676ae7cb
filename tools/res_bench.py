#!/usr/bin/env python
"""Per-shape comparison of the register-resident single-pass BatchNorm-fused kernels with the 3-launch form
(in-situ per-dispatch events).  GPU box only:  python tools/res_bench.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import _lib                                   # noqa: E402
from deepipr_amd.passport_ops import kernels as K              # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(N, 64, 32, 32), (N, 128, 16, 16), (N, 256, 8, 8), (N, 512, 4, 4), (4 * N, 512, 8, 8)]
dev = torch.device('cuda:0')


def run(shape, resident, sync=True, reps=40):
    n, c, h, w = shape
    x = torch.randn(shape, device=dev)
    dy = torch.randn(shape, device=dev)
    g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    _lib.set_resident(resident)
    K.set_user_sync(sync)

    def once():
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
    _lib.set_resident(True)
    K.set_user_sync(True)
    fwd = sum(1000.0 * prof[k][0] / reps for k in ('bn_res_fwd', 'bn_stats', 'gamma_beta_fwd', 'bn_affine_fwd'))
    bwd = sum(1000.0 * prof[k][0] / reps for k in ('bn_res_bwd', 'bn_bwd_reduce', 'passport_bwd_finish', 'bn_affine_bwd'))
    return fwd, bwd


for shape in SHAPES:
    mb = 4 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
    r = run(shape, True)
    ns = run(shape, True, sync=False)
    t = run(shape, False)
    print('%-20s %6.1f MB | resident fwd %6.2f us (%4.0f GB/s) bwd %6.2f us (%4.0f GB/s) | no-sync fwd %6.2f bwd %6.2f'
          ' | 3-launch fwd %6.2f bwd %6.2f' % (shape, mb, r[0], 2 * mb / r[0] * 1e3 / 1e3, r[1], 3 * mb / r[1] * 1e3 / 1e3,
                                             ns[0], ns[1], t[0], t[1]))
print('sync timeouts:', K.sync_timeouts())
