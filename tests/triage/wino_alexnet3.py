"""Triage: which layer's ReLU / pool pattern differs between product and float64 oracle in the gated whole-net AlexNet case."""
import os
import sys

import torch
from _pytest.monkeypatch import MonkeyPatch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import torch_ref                                 # noqa: E402
from tests import test_models_gpu as T                      # noqa: E402

outs = {'p': {}, 'r': {}}
orig = T._whole_net_pair


def pair(*a):
    prod, ref, x, y = orig(*a)
    for side, net in (('p', prod), ('r', ref)):
        for n, m in net.named_modules():
            if n.startswith('features.') and n.count('.') == 1:
                m.register_forward_hook(lambda mod, i, o, n=n, side=side: outs[side].__setitem__(
                    n, (o[0] if isinstance(o, tuple) else o).detach().double().cpu()))
    return prod, ref, x, y


T._whole_net_pair = pair
mp = MonkeyPatch()
try:
    T.test_whole_net_backward_within_1e4_with_relu_kinks_gated('alexnet_v1', mp)
    print('PASSED')
except AssertionError as e:
    print('FAILED', str(e)[:300])
finally:
    mp.undo()
for n in sorted(outs['r'], key=lambda s: int(s.split('.')[1])):
    p, r = outs['p'][n], outs['r'][n]
    if p.shape != r.shape:
        print(n, 'shape mismatch', p.shape, r.shape)
        continue
    d = (p - r).abs()
    flips = ((p > 0) != (r > 0))
    print('%-12s max|diff| %.3g (scale %.3g)  sign-pattern mismatches %d of %d;  at mismatches ref values: %s' % (
        n, float(d.max()), float(r.abs().max()), int(flips.sum()), flips.numel(),
        [float('%.3g' % v) for v in r[flips][:6].tolist()] + [float('%.3g' % v) for v in p[flips][:6].tolist()]))
