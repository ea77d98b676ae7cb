"""Triage: run the whole-net AlexNet case with every own convolution call checked against float64 on its actual operands."""
import os
import sys

import torch
from _pytest.monkeypatch import MonkeyPatch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepipr_amd import passport_ops as P                    # noqa: E402
from tests import test_models_gpu as T                      # noqa: E402

K = P.kernels
of, od = K.conv_fwd, K.conv_dgrad


def fwd(x, w, st, pad):
    y = of(x, w, st, pad)
    if y is not None:
        r = torch.ops.aten.convolution(x.double(), w.double(), None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1)
        print('fwd  ', tuple(x.shape), tuple(w.shape), 'contig', x.is_contiguous(), w.is_contiguous(), 'err/scale %.2e' % (float((y.double() - r).abs().max()) / float(r.abs().max())),
              'wino' if K.conv_is_winograd(x.shape[0], x.shape[1], w.shape[0], x.shape[2], x.shape[3], w.shape[2], st, pad, 0) else 'direct')
    return y


def dgrad(dy, w, xs, st, pad):
    dx = od(dy, w, xs, st, pad)
    if dx is not None:
        xz = torch.zeros(xs, device=dy.device, dtype=torch.float64)
        r = torch.ops.aten.convolution_backward(dy.double(), xz, w.double(), None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1,
                                                [True, False, False])[0]
        print('dgrad', tuple(dy.shape), tuple(w.shape), 'contig', dy.is_contiguous(), w.is_contiguous(), 'err/scale %.2e' % (float((dx.double() - r).abs().max()) / float(r.abs().max())),
              'dy absmax %.3g' % float(dy.abs().max()), 'nan' if bool(torch.isnan(dx).any()) else '')
    return dx


K.conv_fwd, K.conv_dgrad = fwd, dgrad
mp = MonkeyPatch()
try:
    T.test_whole_net_backward_within_1e4_with_relu_kinks_gated('alexnet_v1', mp)
    print('PASSED')
except AssertionError as e:
    print('FAILED', str(e)[:1500])
finally:
    mp.undo()
