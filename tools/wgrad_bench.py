#!/usr/bin/env python
"""Weight-gradient kernel (deepipr_conv_wgrad) against the vendor library, per convolution shape.

    python tools/wgrad_bench.py [--batch 128] [--reps 30] [--json out.json]

Correctness: against aten::convolution_backward in float64 on the GPU (max error over |dW| scale).
Time: HIP events around `reps` back-to-back calls, both sides -- the library's figure therefore INCLUDES its layout
transposes and zero fill, as a train step pays them.  TFLOP/s = 2 * Co * Ci * 9 * N * H * W / time."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402

# (Ci, Co, H = W of the input, stride): ResNet18's 3x3 convolutions
SHAPES = [(3, 64, 32, 1), (64, 64, 32, 1), (128, 128, 16, 1), (256, 256, 8, 1), (512, 512, 4, 1),
          (64, 128, 32, 2), (128, 256, 16, 2), (256, 512, 8, 2),
          (64, 128, 32, 2, 1), (128, 256, 16, 2, 1), (256, 512, 8, 2, 1)]      # ... and its 1x1 stride-2 shortcuts


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
    ap.add_argument('--no-find', action='store_true')
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = not args.no_find
    dev = torch.device('cuda:0')
    out = []
    for shape in SHAPES:
        ci, co, hw, st = shape[:4]
        k = shape[4] if len(shape) > 4 else 3
        pad = k // 2
        n = args.batch
        g = torch.Generator(device='cpu').manual_seed(ci + hw)
        x = torch.randn(n, ci, hw, hw, generator=g).to(dev)
        dy = torch.randn(n, co, hw // st, hw // st, generator=g).to(dev)
        w = torch.randn(co, ci, k, k, generator=g).to(dev)
        ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [st, st], [pad, pad], [1, 1],
                                                  False, [0, 0], 1, [False, True, False])[1]
        lib = lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1,
                                                          [False, True, False])[1]
        mine = lambda: K.conv_wgrad(x, dy, w.shape, st, pad)
        got = mine()
        assert got is not None, 'shape outside the kernel'
        torch.cuda.synchronize()
        scale = float(ref.abs().max())
        err = float((got.double() - ref).abs().max()) / scale
        err_lib = float((lib().double() - ref).abs().max()) / scale
        again = mine()
        bit = bool(torch.equal(got, again))
        t_mine, t_lib = timeit(mine, args.reps), timeit(lib, args.reps)
        # the same call on the DIRECT fp32-MFMA kernel (the default is Winograd F(3x3, 2x2) for the 3x3 stride-1 shapes, or
        # bf16x3 when DEEPIPR_CONV_ARITH selects it)
        before = K.set_conv_arith('fp32')
        algo = K.set_conv_algo('direct')
        got32 = mine()
        torch.cuda.synchronize()
        err32 = float((got32.double() - ref).abs().max()) / scale
        rms = lambda a: float(((a.double() - ref) ** 2).mean().sqrt()) / scale
        t_32 = timeit(mine, args.reps)
        K.set_conv_arith(before)
        K.set_conv_algo(algo)
        split = not torch.equal(got32, got)
        flops = 2.0 * co * ci * k * k * n * (hw // st) ** 2
        rec = {'Ci': ci, 'Co': co, 'HW': hw, 'k': k, 'stride': st, 'N': n, 'us': round(t_mine, 1), 'us_library': round(t_lib, 1),
               'TFLOPs': round(flops / t_mine / 1e6, 1), 'TFLOPs_library': round(flops / t_lib / 1e6, 1),
               'err_over_scale': err, 'err_library': err_lib, 'bit_reproducible': bit,
               'arith': K.conv_arith() if split else 'fp32', 'algo': K.conv_algo() if split else 'direct',
               'us_fp32_mfma': round(t_32, 1), 'err_fp32_mfma': err32,
               'rms_err': rms(got), 'rms_err_fp32_mfma': rms(got32), 'rms_err_library': rms(lib())}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
