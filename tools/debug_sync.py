import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import patterns
from oracle.cases import ALPHA, resnet18_config
from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
from deepipr_amd.models.resnet_passport_private import ResNet18Private
from deepipr_amd import passport_ops as PO
import torch.nn.functional as F
DEV = 'cuda:0'
torch.backends.cudnn.benchmark = False
torch.backends.cudnn.deterministic = True
cfg = resnet18_config()
x, y = patterns.batch(64, 3, 32, 32, 100)
x, y = x.to(DEV), y.to(DEV)
SYNC = False
K = PO.kernels
for name in ('passport_bn_fwd', 'passport_bn_bwd'):
    orig = getattr(K, name)
    def wrap(orig):
        def f(*a, **k):
            if SYNC: torch.cuda.synchronize()
            out = orig(*a, **k)
            if SYNC: torch.cuda.synchronize()
            return out
        return f
    setattr(K, name, wrap(orig))


def run(fuse, inds):
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
    loss = 0
    for ind in inds:
        loss = loss + F.cross_entropy(m(x, ind=ind), y)
    loss.backward()
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


def worst(B, C):
    w = sorted(((float((B[k] - C[k]).abs().max()) / (float(B[k].abs().max()) + 1e-12), k) for k in B), reverse=True)[:3]
    return ' '.join('%s %.1e' % (k, d) for d, k in w)


B = run(False, (0,))
for s in (False, True, False):
    SYNC = s
    C = run(True, (0,))
    print('sync=%s' % s, worst(B, C))
os.environ['AMD_SERIALIZE_KERNEL'] = '3'
