#!/usr/bin/env python
"""The ImageNet stem (Conv 3 -> 64, 7x7 / 2, pad 3, 224 x 224 images; models/resnet_passport.py:94-98): this library's forward kernel
(deepipr_conv_stem7.inc) against the vendor library's convolution, HIP events around back-to-back calls.

    python tools/stem7_bench.py [--batch 256] [--reps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402


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
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    x = torch.randn(a.batch, 3, 224, 224, device=dev)
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
    y = K.conv_fwd(x, w, 2, 3)
    ref = torch.nn.functional.conv2d(x, w, None, 2, 3)
    flops = 2.0 * 64 * 147 * a.batch * 112 * 112
    us = timeit(lambda: K.conv_fwd(x, w, 2, 3), a.reps)
    us_lib = timeit(lambda: torch.nn.functional.conv2d(x, w, None, 2, 3), a.reps)
    dy = torch.randn_like(y)
    dw = K.conv_wgrad(x, dy, (64, 3, 7, 7), 2, 3)

    def lib_wgrad():
        return torch.ops.aten.convolution_backward(dy, x, w, None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1, [False, True, False])[1]
    dw_lib = lib_wgrad()
    wus = timeit(lambda: K.conv_wgrad(x, dy, (64, 3, 7, 7), 2, 3), a.reps)
    wus_lib = timeit(lib_wgrad, a.reps)
    print(json.dumps({'batch': a.batch, 'wgrad_us': round(wus, 1), 'wgrad_useful_TFLOPs': round(flops / wus / 1e6, 1),
                      'wgrad_GBs': round(4.0 * (x.numel() + dy.numel()) / wus / 1e3, 1), 'wgrad_library_us': round(wus_lib, 1),
                      'wgrad_max_diff_vs_library_of_scale': float((dw - dw_lib).abs().max() / dw_lib.abs().max())}))
    print(json.dumps({'batch': a.batch, 'fwd_us': round(us, 1), 'useful_TFLOPs': round(flops / us / 1e6, 1),
                      'executed_TFLOPs': round(flops * (168 / 147) * (32 / 28) / us / 1e6, 1),
                      'GBs': round(4.0 * (x.numel() + y.numel()) / us / 1e3, 1), 'library_us': round(us_lib, 1),
                      'max_diff_vs_library': float((y - ref).abs().max())}))


if __name__ == '__main__':
    main()
