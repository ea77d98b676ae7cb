"""Kernel names MIOpen runs for ONE convolution's forward / backward in pinned mode (cudnn.benchmark = False,
deterministic = True), via torch.profiler -- to name the solver behind a nondeterministic result.
    python tools/conv_kernels.py N Cin Cout H k stride pad"""
import json
import sys

import torch
from torch.profiler import ProfilerActivity, profile

n, ci, co, h, k, s, p = [int(v) for v in sys.argv[1:8]]
torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
dev = 'cuda:0'
g = torch.Generator().manual_seed(7)
x = torch.randn(n, ci, h, h, generator=g).to(dev)
w = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
ho = (h + 2 * p - k) // s + 1
dy = torch.randn(n, co, ho, ho, generator=g).to(dev)


def run():
    y = torch.ops.aten.convolution(x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1)
    dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1, [True, True, False])
    return y, dx, dw


first = run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    run()
    torch.cuda.synchronize()
names = [e.name for e in prof.events() if 'cuda' in str(getattr(e, 'device_type', '')).lower()]
diff = {'y': 0, 'dx': 0, 'dw': 0}
for _ in range(50):
    for key, a, b in zip(('y', 'dx', 'dw'), run(), first):
        diff[key] += int(not torch.equal(a, b))
print(json.dumps({'conv': sys.argv[1:8], 'kernels': names, 'repetitions_differing_of_50': diff}))
