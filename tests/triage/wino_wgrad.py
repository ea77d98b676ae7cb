"""Triage of k_conv_wino_wgrad: small shapes, per-tap / per-block error, determinism."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepipr_amd.passport_ops import kernels as K           # noqa: E402

dev = 'cuda:0'


def ref(x, dy, wshape):
    w = torch.zeros(wshape, dtype=torch.float64, device=x.device)
    return torch.ops.aten.convolution_backward(dy.double(), x.double(), w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]


for (n, ci, co, h, w) in [(1, 32, 64, 2, 32), (1, 32, 64, 4, 32), (2, 32, 64, 32, 32), (1, 32, 64, 8, 8), (4, 32, 64, 4, 4), (128, 64, 64, 32, 32)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, ci, h, w, generator=g).to(dev)
    dy = torch.randn(n, co, h, w, generator=g).to(dev)
    r = ref(x, dy, (co, ci, 3, 3))
    a = K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1)
    b = K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1)
    torch.cuda.synchronize()
    e = (a.double() - r).abs()
    sc = float(r.abs().max())
    print((n, ci, co, h, w), 'algo', K.conv_algo(), 'err/scale %.2e' % (float(e.max()) / sc), 'repeat equal', bool(torch.equal(a, b)),
          'per tap', [float('%.1e' % (float(e[:, :, i // 3, i % 3].max()) / sc)) for i in range(9)],
          'co<32 %.1e co>=32 %.1e' % (float(e[:32].max()) / sc, float(e[32:].max()) / sc),
          'nan', bool(torch.isnan(a).any()))
    # ratio got / ref where ref is large
    m = r.abs() > 0.5 * sc
    print('    got/ref at large entries: min %.3f max %.3f' % (float((a.double()[m] / r[m]).min()), float((a.double()[m] / r[m]).max())))
