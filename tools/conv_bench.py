#!/usr/bin/env python
"""Forward / backward-data convolution kernels (deepipr_conv_fwd, deepipr_conv_dgrad) against the vendor library, per shape.

    python tools/conv_bench.py [--batch 128] [--reps 30] [--json out.json]

Correctness against ATen in float64 on the GPU; time by HIP events around back-to-back calls (the library's figure includes
its layout shims, as a train step pays them)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402

# (Ci, Co, H = W of the input, k, stride): the convolutions of ResNet18 (CIFAR) behind the stem
SHAPES = [(64, 64, 32, 3, 1), (128, 128, 16, 3, 1), (256, 256, 8, 3, 1), (512, 512, 4, 3, 1),
          (64, 128, 32, 3, 2), (128, 256, 16, 3, 2), (256, 512, 8, 3, 2),
          (64, 128, 32, 1, 2), (128, 256, 16, 1, 2), (256, 512, 8, 1, 2)]


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1000.0 * a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--json', default=None)
    ap.add_argument('--imagenet', action='store_true', help='the 3x3 stride-1 shapes of ResNet18 / ResNet50 at 224 x 224')
    ap.add_argument('--algo', default=None, choices=['direct', 'winograd'], help='3x3 stride-1 algorithm (default: the library default)')
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    if args.algo:
        K.set_conv_algo(args.algo)
    dev = torch.device('cuda:0')
    out = []
    shapes = [(64, 64, 56, 3, 1), (128, 128, 28, 3, 1), (256, 256, 14, 3, 1), (512, 512, 7, 3, 1)] if args.imagenet else SHAPES
    for ci, co, hw, k, st in shapes:
        n, pad = args.batch, k // 2
        g = torch.Generator(device='cpu').manual_seed(ci + hw + k)
        x = torch.randn(n, ci, hw, hw, generator=g).to(dev)
        w = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
        dy = torch.randn(n, co, hw // st, hw // st, generator=g).to(dev)
        conv = lambda a, b: torch.ops.aten.convolution(a, b, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1)
        y_ref = conv(x.double(), w.double())
        rec = {'Ci': ci, 'Co': co, 'HW': hw, 'k': k, 'stride': st, 'N': n,
               'algo': 'winograd' if K.conv_is_winograd(n, ci, co, hw, hw, k, st, pad, 0) else 'direct'}
        flops = 2.0 * co * ci * k * k * n * (hw // st) ** 2
        y = K.conv_fwd(x, w, st, pad)
        if y is not None:
            torch.cuda.synchronize()
            rec['fwd_err'] = float((y.double() - y_ref).abs().max()) / float(y_ref.abs().max())
            rec['fwd_err_lib'] = float((conv(x, w).double() - y_ref).abs().max()) / float(y_ref.abs().max())
            t, tl = timeit(lambda: K.conv_fwd(x, w, st, pad), args.reps), timeit(lambda: conv(x, w), args.reps)
            rec.update(fwd_us=round(t, 1), fwd_us_lib=round(tl, 1), fwd_TF=round(flops / t / 1e6, 1))
        bwd = lambda d, a, b: torch.ops.aten.convolution_backward(d, a, b, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1,
                                                                  [True, False, False])[0]
        dx = K.conv_dgrad(dy, w, x.shape, st, pad)
        if dx is not None:
            dx_ref = bwd(dy.double(), x.double(), w.double())
            torch.cuda.synchronize()
            rec['dgrad_err'] = float((dx.double() - dx_ref).abs().max()) / float(dx_ref.abs().max())
            t, tl = timeit(lambda: K.conv_dgrad(dy, w, x.shape, st, pad), args.reps), timeit(lambda: bwd(dy, x, w), args.reps)
            rec.update(dgrad_us=round(t, 1), dgrad_us_lib=round(tl, 1), dgrad_TF=round(flops / t / 1e6, 1))
        print(json.dumps(rec), flush=True)
        out.append(rec)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
