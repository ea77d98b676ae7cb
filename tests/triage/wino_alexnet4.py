"""Triage: the same AlexNet V1 product net run under DEEPIPR conv algo winograd vs direct: per-layer output / gradient diffs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepipr_amd.passport_ops import kernels as K           # noqa: E402
from tests import test_models_gpu as T                      # noqa: E402

prod, ref, x, y = T._whole_net_pair('alexnet', False, 64, 10, 'bn')
xg, yg = x.to('cuda:0'), y.to('cuda:0')
state = {k: v.clone() for k, v in prod.state_dict().items()}
res = {}
for algo in ('winograd', 'direct', 'winograd'):
    K.set_conv_algo(algo)
    prod.load_state_dict(state)
    prod.zero_grad(set_to_none=True)
    outs = {}
    hooks = [m.register_forward_hook(lambda mod, i, o, n=n: outs.__setitem__(n, (o[0] if isinstance(o, tuple) else o).detach().clone()))
             for n, m in prod.named_modules() if n.startswith('features.') and n.count('.') == 1]
    out = prod(xg)
    loss = torch.nn.functional.cross_entropy(out, yg) + sum(m.sign_loss.loss for m in prod.modules()
                                                            if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
    loss.backward()
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    grads = {n: p.grad.detach().clone() for n, p in prod.named_parameters()}
    if algo in res:
        a = res[algo]
        print('repeat of %s: bitwise equal outputs %s grads %s' % (algo, all(torch.equal(a[0][k], outs[k]) for k in outs),
                                                                 all(torch.equal(a[1][k], grads[k]) for k in grads)))
    res[algo] = (outs, grads)
ow, gw = res['winograd']
od, gd = res['direct']
for n in sorted(ow, key=lambda s: int(s.split('.')[1])):
    d = (ow[n] - od[n]).abs()
    i = int(d.argmax())
    print('%-11s out max|wino - direct| %.3g (scale %.3g) at flat %d: wino %.6g direct %.6g;  elements > 1e-4: %d' % (
        n, float(d.max()), float(od[n].abs().max()), i, float(ow[n].flatten()[i]), float(od[n].flatten()[i]), int((d > 1e-4).sum())))
for n in gw:
    d = (gw[n] - gd[n]).abs()
    print('%-26s grad rel diff %.3g' % (n, float(d.max()) / (float(gd[n].abs().max()) + 1e-30)))
