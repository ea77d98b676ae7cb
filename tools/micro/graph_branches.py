#!/usr/bin/env python
"""Does a replayed hipGraph run independent branches concurrently on this stack (ROCm 7.2 / torch 2.10)?  Twenty small-grid
kernels (a convolution that fills < 1/4 of the chip) captured (a) on one stream, (b) forked over two streams, (c) eager on two
streams; wall time per pass."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402

dev = torch.device('cuda:0')
x = [torch.randn(2, 512, 4, 4, device=dev) for _ in range(2)]
w = [torch.randn(512, 512, 3, 3, device=dev) * 0.05 for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def work(i, n):
    for _ in range(n):
        K.conv_fwd(x[i], w[i], 1, 1)


def forked():
    main = torch.cuda.current_stream()
    s2.wait_stream(main)
    with torch.cuda.stream(s2):
        work(1, 10)
    work(0, 10)
    main.wait_stream(s2)


def timed(fn, reps=50):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


with torch.cuda.stream(s1):
    work(0, 2); work(1, 2); forked()
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    work(0, 10); work(1, 10)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s1):
    forked()
print('graph, one stream   : %.1f us' % timed(g1.replay))
print('graph, two branches : %.1f us' % timed(g2.replay))
with torch.cuda.stream(s1):
    print('eager, one stream   : %.1f us' % timed(lambda: (work(0, 10), work(1, 10)), 20))
    print('eager, two streams  : %.1f us' % timed(forked, 20))
