#!/usr/bin/env python
"""The 1x1 stride-1 convolutions of ResNet50 at ImageNet geometry (BASELINE config 5), per shape: this library's weight
gradient (deepipr_conv_1x1.inc) against the vendor library's (which wraps its NHWC kernel in layout transposes).

    python tools/conv1x1_bench.py [--batch 256] [--reps 10] [--json out.json]

Correctness against aten::convolution_backward in float64 on the GPU (skipped with --no-check); time: HIP events around `reps`
back-to-back calls, both sides.  TFLOP/s = 2 Co Ci N HW / time; GB/s = 4 (|x| + |dy|) / time (the layer-1 shapes sit at the
ridge: 25.6 FLOP per byte)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402

# (Ci, Co, H = W): the distinct 1x1 stride-1 convolutions of ResNet50 (models/resnet_normal.py:30-49) and how many there are
SHAPES = [(64, 64, 56, 1), (64, 256, 56, 4), (256, 64, 56, 2), (256, 128, 56, 1),
          (128, 512, 28, 4), (512, 128, 28, 3), (512, 256, 28, 1),
          (256, 1024, 14, 6), (1024, 256, 14, 5), (1024, 512, 14, 1),
          (512, 2048, 7, 3), (2048, 512, 7, 2)]


def timeit(fn, reps):
    for _ in range(2):
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
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--json', default=None)
    ap.add_argument('--no-check', action='store_true')
    ap.add_argument('--no-library', action='store_true')
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device('cuda:0')
    out, tot, tot_lib, fd = [], 0.0, 0.0, [0.0, 0.0]
    for ci, co, hw, count in SHAPES:
        n = args.batch
        g = torch.Generator(device='cpu').manual_seed(ci + hw)
        x = torch.randn(n, ci, hw, hw, generator=g).to(dev)
        dy = torch.randn(n, co, hw, hw, generator=g).to(dev)
        w = torch.randn(co, ci, 1, 1, generator=g).to(dev)
        lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1,
                                                          [False, True, False])[1]
        mine = lambda: K.conv_wgrad(x, dy, w.shape, 1, 0)
        got = mine()
        assert got is not None, 'shape outside the kernel'
        rec = {'Ci': ci, 'Co': co, 'HW': hw, 'N': n, 'count': count}
        if not args.no_check:
            ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [0, 0], [1, 1],
                                                      False, [0, 0], 1, [False, True, False])[1]
            scale = float(ref.abs().max())
            rec['err_over_scale'] = float((got.double() - ref).abs().max()) / scale
            rec['err_library'] = float((lib().double() - ref).abs().max()) / scale
            rec['bit_reproducible'] = bool(torch.equal(got, mine()))
            del ref
        t = timeit(mine, args.reps)
        flops = 2.0 * co * ci * n * hw * hw
        byts = 4.0 * n * hw * hw * (ci + co)
        rec.update({'us': round(t, 1), 'TFLOPs': round(flops / t / 1e6, 1), 'GBs': round(byts / t / 1e3, 0)})
        tot += count * t
        if not args.no_library:
            tl = timeit(lib, args.reps)
            rec.update({'us_library': round(tl, 1), 'TFLOPs_library': round(flops / tl / 1e6, 1)})
            tot_lib += count * tl
        # forward / backward-data: this library's NCHW GEMM (k_conv1x1_gemm) against the BLAS library's batched GEMM (the route
        # passport_ops._conv_fwd / _conv_dgrad take without it)
        fwd_own = lambda: K.conv_fwd(x, w, 1, 0)
        dg_own = lambda: K.conv_dgrad(dy, w, tuple(x.shape), 1, 0)
        fwd_blas = lambda: torch.bmm(w.view(1, co, ci).expand(n, -1, -1), x.view(n, ci, hw * hw))
        dg_blas = lambda: torch.bmm(w.view(1, co, ci).transpose(1, 2).expand(n, -1, -1), dy.view(n, co, hw * hw))
        if fwd_own() is not None:
            tf, tb, tfb, tbb = timeit(fwd_own, args.reps), timeit(dg_own, args.reps), timeit(fwd_blas, args.reps), timeit(dg_blas, args.reps)
            rec.update({'fwd_us': round(tf, 1), 'fwd_TFLOPs': round(flops / tf / 1e6, 1), 'fwd_us_blas': round(tfb, 1),
                        'dgrad_us': round(tb, 1), 'dgrad_TFLOPs': round(flops / tb / 1e6, 1), 'dgrad_us_blas': round(tbb, 1)})
            fd[0] += count * (tf + tb)
            fd[1] += count * (tfb + tbb)
        print(json.dumps(rec), flush=True)
        out.append(rec)
    summary = {'summary': 'all 33 launches of a ResNet50 step', 'ms': round(tot / 1e3, 2), 'ms_library': round(tot_lib / 1e3, 2),
               'fwd_plus_dgrad_ms': round(fd[0] / 1e3, 2), 'fwd_plus_dgrad_ms_blas': round(fd[1] / 1e3, 2)}
    print(json.dumps(summary), flush=True)
    out.append(summary)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
