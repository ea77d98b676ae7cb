"""One pinned-MIOpen (cudnn.benchmark = False, deterministic = True) train step of the bit-identity tests' nets, for a
rocprofv3 --kernel-trace --stats of WHICH vendor kernels that mode runs:  python tools/pinned_step.py [--private]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_parity_gpu as T                     # noqa: E402

private = '--private' in sys.argv
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
n, ncls = (64, 100) if private else (128, 10)
prod, _ref, x, y = T._fullsize_pair(private, n, ncls)
x, y = x.to(T.DEV), y.to(T.DEV)
ce = torch.nn.functional.cross_entropy
for _ in range(3):
    prod.zero_grad(set_to_none=True)
    if private:
        outs = [prod(x, ind=0), prod(x, ind=1)]
        loss = ce(outs[0], y) + ce(outs[1], y) + sum(m.sign_loss_private.loss for m in prod.modules()
                                                     if hasattr(m, 'sign_loss_private'))
    else:
        loss = ce(prod(x), y) + sum(m.sign_loss.loss for m in prod.modules()
                                    if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
    loss.backward()
torch.cuda.synchronize()
print('ok')
