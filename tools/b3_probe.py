#!/usr/bin/env python
"""Time the bf16x3 weight-gradient kernel alone (HIP events, back to back) for the ResNet18 stride-1 shapes."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K
dev = torch.device('cuda:0')
out = {}
for ci, hw in ((64, 32), (128, 16), (256, 8)):
    x = torch.randn(128, ci, hw, hw, device=dev); dy = torch.randn(128, ci, hw, hw, device=dev)
    f = lambda: K.conv_wgrad(x, dy, (ci, ci, 3, 3), 1, 1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50): f()
    b.record(); torch.cuda.synchronize()
    out['%dx%d' % (ci, hw)] = round(1000 * a.elapsed_time(b) / 50, 1)
if int(os.environ.get('DEEPIPR_B3_DBG', '0')) & 16:
    torch.cuda.synchronize()
    for key, buf in K._arena.items():
        if isinstance(key[1], tuple) and key[1][0] == 'wgrad':
            v = buf[:40].view(torch.int64).tolist()
            out['cycles'] = dict(loader_work=v[0], loader_barrier=v[1], mfma_work=v[2], mfma_barrier=v[3], chunks=v[4])
print(os.environ.get('DEEPIPR_B3_DBG', '0'), json.dumps(out))
