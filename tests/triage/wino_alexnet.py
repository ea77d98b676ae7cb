"""Triage: per-convolution error of the AlexNet V1 whole-net case (pattern-filled weights) under the Winograd kernels, the
direct kernels and the vendor library, forward and backward-data, against float64 ATen on the SAME operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import test_models_gpu as T                      # noqa: E402
from deepipr_amd.passport_ops import kernels as K           # noqa: E402

prod, ref, x, y = T._whole_net_pair('alexnet', False, 64, 10, 'bn')
caught = []
for name, m in prod.named_modules():
    if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1):
        m.register_forward_pre_hook(lambda mod, inp, name=name: caught.append((name, mod, inp[0].detach().clone())))
        print('hooked', name, tuple(m.weight.shape))
# the fused layers call the conv functionally: capture through the weights instead
convs = [(n, m) for n, m in prod.named_modules() if isinstance(m, torch.nn.Conv2d)]
acts = {}
hooks = []
for n, m in prod.named_modules():
    if n.startswith('features.') and n.count('.') == 1:
        hooks.append(m.register_forward_hook(lambda mod, i, o, n=n: acts.__setitem__(n, (i[0].detach().clone(), o))))
with torch.no_grad():
    prod(x.to('cuda:0'))
for n, (i, o) in acts.items():
    print(n, tuple(i.shape), 'in: min %.3g max %.3g mean %.3g' % (float(i.min()), float(i.max()), float(i.mean())))
conv64 = lambda a, b: torch.ops.aten.convolution(a.double(), b.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
for n, m in convs:
    if m.kernel_size != (3, 3):
        continue
    layer = n.rsplit('.', 1)[0]
    if layer not in acts:
        continue
    xin = acts[layer][0].contiguous()
    w = m.weight.detach().contiguous()
    ref64 = conv64(xin, w)
    scale = float(ref64.abs().max())
    print(layer, 'w: absmax %.3g' % float(w.abs().max()), 'out scale %.3g rms %.3g' % (scale, float(ref64.pow(2).mean().sqrt())))
    dy = torch.randn_like(ref64, dtype=torch.float32)
    dref = torch.ops.aten.convolution_backward(dy.double(), xin.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                               [True, False, False])[0]
    for algo in ('winograd', 'direct'):
        K.set_conv_algo(algo)
        yk = K.conv_fwd(xin, w, 1, 1)
        dk = K.conv_dgrad(dy, w, xin.shape, 1, 1)
        if yk is not None:
            e = (yk.double() - ref64).abs()
            print('   %-8s fwd max err %.3g (%.2g of scale) rms %.3g | dgrad max err %.3g (%.2g of scale)' % (
                algo, float(e.max()), float(e.max()) / scale, float(e.pow(2).mean().sqrt()),
                float((dk.double() - dref).abs().max()), float((dk.double() - dref).abs().max()) / float(dref.abs().max())))
    yl = torch.ops.aten.convolution(xin, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
    e = (yl.double() - ref64).abs()
    print('   library  fwd max err %.3g (%.2g of scale) rms %.3g' % (float(e.max()), float(e.max()) / scale, float(e.pow(2).mean().sqrt())))
