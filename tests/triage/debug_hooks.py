import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import patterns
from oracle.cases import ALPHA, resnet18_config
from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
from deepipr_amd.models.resnet_passport_private import ResNet18Private
import torch.nn.functional as F
DEV = 'cuda:0'
torch.backends.cudnn.benchmark = False
torch.backends.cudnn.deterministic = True
cfg = resnet18_config()
x, y = patterns.batch(64, 3, 32, 32, 100)
x, y = x.to(DEV), y.to(DEV)


def run(fuse):
    torch.manual_seed(0); np.random.seed(0)
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random', 'sl_ratio': ALPHA})
    m = ResNet18Private(num_classes=100, passport_kwargs=kw).to(DEV)
    m.train()
    with torch.no_grad():
        m(x)
    patterns.fill_state(m)
    rec = {}
    for name, mod in m.named_modules():
        if hasattr(mod, 'fuse_norm'):
            mod.fuse_norm = fuse
            def fh(mod, inp, out, name=name):
                rec['y/' + name] = out.detach().clone()
                rec['xin/' + name] = inp[0].detach().clone()
                out.register_hook(lambda g, name=name: rec.__setitem__('dy/' + name, g.detach().clone()))
                if inp[0].requires_grad:
                    inp[0].register_hook(lambda g, name=name: rec.__setitem__('dxin/' + name, g.detach().clone()))
            mod.register_forward_hook(fh)
    loss = F.cross_entropy(m(x, ind=0), y)
    loss.backward()
    for n, p in m.named_parameters():
        if p.grad is not None and n.startswith('layer4'):
            rec['grad/' + n] = p.grad.clone()
    return rec


B, C = run(False), run(True)
for k in sorted(B, key=lambda s: (s.split('/')[1], s.split('/')[0])):
    d = float((B[k] - C[k]).abs().max()); s = float(B[k].abs().max()) + 1e-12
    nz = int(((B[k] - C[k]).abs() > 1e-3 * s).sum())
    print('%-44s rel %.2e  (n > 1e-3: %d of %d)' % (k, d / s, nz, B[k].numel()))

# ---- which path is closer to float64 ground truth on the last passport layer? ----
print('--- fp64 ground truth for layer4.1.convbn_2 (public branch) ---')


def layer_truth(fuse):
    torch.manual_seed(0); np.random.seed(0)
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random', 'sl_ratio': ALPHA})
    m = ResNet18Private(num_classes=100, passport_kwargs=kw).to(DEV)
    m.train()
    with torch.no_grad():
        m(x)
    patterns.fill_state(m)
    for mod in m.modules():
        if hasattr(mod, 'fuse_norm'):
            mod.fuse_norm = fuse
    blk = m.layer4[1].convbn_2
    rec = {}
    def h1(mod, i, o):
        rec['x'] = o.detach().clone()
        o.register_hook(lambda g: rec.__setitem__('dx', g.detach().clone()))

    def h2(mod, i, o):
        o.register_hook(lambda g: rec.__setitem__('dy', g.detach().clone()))
    blk.conv.register_forward_hook(h1)
    blk.register_forward_hook(h2)
    F.cross_entropy(m(x, ind=0), y).backward()
    return rec, blk


for fuse in (False, True):
    rec, blk = layer_truth(fuse)
    X, DY = rec['x'].double(), rec['dy'].double()
    g, b = blk.scale.detach().double().view(1, -1, 1, 1), blk.bias.detach().double().view(1, -1, 1, 1)
    mu = X.mean(dim=(0, 2, 3), keepdim=True); var = X.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    istd = 1 / torch.sqrt(var + 1e-5)
    xh = (X - mu) * istd
    dz = torch.where(g * xh + b > 0, DY, torch.zeros_like(DY))
    M = X.numel() // X.shape[1]
    dx = g * istd * (dz - dz.mean(dim=(0, 2, 3), keepdim=True) - xh * (dz * xh).mean(dim=(0, 2, 3), keepdim=True))
    err = (rec['dx'].double() - dx).abs()
    s = float(dx.abs().max())
    per_c = err.amax(dim=(0, 2, 3)) / s
    top = torch.topk(per_c, 3)
    ratio = (mu.abs().view(-1) / torch.sqrt(var.view(-1) + 1e-5))
    print('fuse=%s: max|dx - fp64| / max|dx| = %.2e ; worst channels %s err %s |mean|/std %s var %s' % (
        fuse, float(err.max()) / s, top.indices.tolist(), ['%.1e' % v for v in top.values.tolist()],
        ['%.1f' % float(ratio[i]) for i in top.indices], ['%.2e' % float(var.view(-1)[i]) for i in top.indices]))

print('--- conv backward (MIOpen) vs fp64 for layer4.1.convbn_2.conv ---')
CAPS = {}
for fuse in (False, True):
    rec, blk = layer_truth(fuse)
    xin = {}
    # recompute: need the conv input; rerun capturing it
    torch.manual_seed(0); np.random.seed(0)
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random', 'sl_ratio': ALPHA})
    m = ResNet18Private(num_classes=100, passport_kwargs=kw).to(DEV)
    m.train()
    with torch.no_grad():
        m(x)
    patterns.fill_state(m)
    for mod in m.modules():
        if hasattr(mod, 'fuse_norm'):
            mod.fuse_norm = fuse
    blk = m.layer4[1].convbn_2
    cap = {}

    def hin(mod, i, o):
        cap['in'] = i[0].detach().clone()
        cap['dout_hook'] = o.register_hook(lambda g: cap.__setitem__('dout', g.detach().clone()))
        i[0].register_hook(lambda g: cap.__setitem__('din', g.detach().clone()))
    blk.conv.register_forward_hook(hin)
    F.cross_entropy(m(x, ind=0), y).backward()
    wgrad = blk.weight.grad.detach().clone()
    xi = cap['in'].double().requires_grad_(True)
    w64 = blk.weight.detach().double().requires_grad_(True)
    out = F.conv2d(xi, w64, None, 1, 1)
    out.backward(cap['dout'].double())
    e_in = float((cap['din'].double() - xi.grad).abs().max() / xi.grad.abs().max())
    e_w = float((wgrad.double() - w64.grad).abs().max() / w64.grad.abs().max())
    print('fuse=%s: MIOpen dgrad vs fp64 %.2e   wgrad vs fp64 %.2e' % (fuse, e_in, e_w))
    CAPS[fuse] = dict(cap, wgrad=wgrad)
for k in ('in', 'dout', 'din', 'wgrad'):
    a, b = CAPS[False][k], CAPS[True][k]
    d = (a - b).abs()
    print('B vs C %-6s rel %.2e  argmax %s  n>1e-3: %d' % (k, float(d.max() / a.abs().max()), np.unravel_index(int(d.argmax()), tuple(a.shape)), int((d > 1e-3 * a.abs().max()).sum())))
d = (CAPS[False]['dout'] - CAPS[True]['dout']).abs().amax(dim=(0, 2, 3))
print('channels with dout diff:', torch.topk(d, 5))
