#!/usr/bin/env python
"""Experiments on how to drive the whole train step (not part of the product): eager vs hipGraph capture,
MIOpen find mode on/off, channels_last.  Prints ms/step per variant."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepipr_amd.experiments.trainer import train_step_v1  # noqa: E402


def timeit(fn, steps=40, warm=10):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(i)
    torch.cuda.synchronize()
    return 1000 * (time.perf_counter() - t0) / steps


def main():
    import argparse
    args = argparse.Namespace(batch=128, classes=10, scheme=1)
    dev = torch.device('cuda:0')
    xs = [torch.randn(128, 3, 32, 32, device=dev) for _ in range(4)]
    ys = [torch.randint(0, 10, (128,), device=dev) for _ in range(4)]
    for bm in (True, False):
        torch.backends.cudnn.benchmark = bm
        model = bench.build_model(args, dev)
        model.train()
        with torch.no_grad():
            model(xs[0])
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        print('eager benchmark=%s: %.3f ms/step' % (bm, timeit(lambda i: train_step_v1(model, opt, xs[i % 4], ys[i % 4]))), flush=True)
    torch.backends.cudnn.benchmark = True
    # fused/foreach optimizer variants
    for kw in ({'foreach': True}, {'fused': True}):
        try:
            model = bench.build_model(args, dev)
            model.train()
            with torch.no_grad():
                model(xs[0])
            opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4, **kw)
            print('eager SGD %s: %.3f ms/step' % (kw, timeit(lambda i: train_step_v1(model, opt, xs[i % 4], ys[i % 4]))), flush=True)
        except Exception as e:
            print('SGD', kw, 'failed:', e)
    # whole-step hipGraph capture (static input buffers)
    try:
        model = bench.build_model(args, dev)
        model.train()
        with torch.no_grad():
            model(xs[0])
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        sx, sy = xs[0].clone(), ys[0].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                train_step_v1(model, opt, sx, sy)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            out = train_step_v1(model, opt, sx, sy)

        def replay(i):
            sx.copy_(xs[i % 4])
            sy.copy_(ys[i % 4])
            g.replay()
        print('hipGraph whole step: %.3f ms/step' % timeit(replay), flush=True)
        print('loss after replays', float(out[0]))
    except Exception as e:
        import traceback
        traceback.print_exc()
        print('graph capture failed:', e)


if __name__ == '__main__':
    main()
