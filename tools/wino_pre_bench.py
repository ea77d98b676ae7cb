#!/usr/bin/env python
"""Winograd forward / backward-data: every workgroup transforms its filters (deepipr_conv_fwd_ws) against the pre-transformed
form (deepipr_conv_wino_transform_multi once, deepipr_conv_fwd_pre / _dgrad_pre), per layer shape of ResNet18, and the
transform launch for all thirteen 3x3 stride-1 weights.  HIP events around back-to-back calls.

    python tools/wino_pre_bench.py [--batch 128] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402
from tools.conv_bench import timeit                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=128)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    dev, n, out = torch.device('cuda:0'), args.batch, []
    for c, hw in [(64, 32), (128, 16), (256, 8), (512, 4), (64, 56), (128, 28), (256, 14), (512, 7)]:
        nn = n if hw <= 32 else max(1, n // 4)
        x = torch.randn(nn, c, hw, hw, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        dy = torch.randn(nn, c, hw, hw, device=dev)
        pre = K.wino_transform([w])[0]
        rec = {'C': c, 'HW': hw, 'N': nn,
               'same_bits': bool(torch.equal(K.conv_fwd(x, w, 1, 1), K.conv_fwd(x, w, 1, 1, pre))
                                 and torch.equal(K.conv_dgrad(dy, w, x.shape, 1, 1), K.conv_dgrad(dy, w, x.shape, 1, 1, pre))),
               'fwd_us': round(timeit(lambda: K.conv_fwd(x, w, 1, 1), 30), 1),
               'fwd_pre_us': round(timeit(lambda: K.conv_fwd(x, w, 1, 1, pre), 30), 1),
               'dgrad_us': round(timeit(lambda: K.conv_dgrad(dy, w, x.shape, 1, 1), 30), 1),
               'dgrad_pre_us': round(timeit(lambda: K.conv_dgrad(dy, w, x.shape, 1, 1, pre), 30), 1),
               'transform_us': round(timeit(lambda: K.wino_transform([w]), 30), 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    ws = [torch.randn(c, c, 3, 3, device=dev) * 0.05 for c in [64] * 4 + [128] * 3 + [256] * 3 + [512] * 3]
    rec = {'resnet18_all_13_weights_us': round(timeit(lambda: K.wino_transform(ws), 30), 1),
           'forward_images_only_us': round(timeit(lambda: K.wino_transform(ws, backward=False), 30), 1),
           'without_layer4_us': round(timeit(lambda: K.wino_transform(ws[:10]), 30), 1)}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
