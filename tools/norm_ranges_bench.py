#!/usr/bin/env python
"""The fused norm kernels on the activation maps of ResNet50 at ImageNet geometry (BASELINE config 5, batch 256): forward and
backward per map, GB/s of the algorithmic bytes (8 / 12 B per element) against the 8 TB/s HBM peak.

    [DEEPIPR_BN_RANGES=0|1] [DEEPIPR_BN_STAGGER=f,b] python tools/norm_ranges_bench.py [--batch 256] [--reps 5] [--json out]

DEEPIPR_BN_RANGES=0: one launch per channel range (round 5); 1 (default): all ranges in one launch.  The library reads both
switches once, so A/B runs are separate processes (tools/norm_ranges_sweep.sh)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd import _lib                                   # noqa: E402
from deepipr_amd.passport_ops import kernels as K              # noqa: E402

# (C, H = W, how many norm layers of the net see this map): the stem, then conv1 / conv2 / conv3 (+ projection) outputs per stage
MAPS = [(64, 112, 1), (64, 56, 6), (256, 56, 4), (128, 56, 1), (128, 28, 7), (512, 28, 5), (256, 28, 1), (256, 14, 11), (1024, 14, 7),
        (512, 14, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--json', default=None)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    out, tot = [], [0.0, 0.0]
    for c, hw, count in MAPS:
        shape = (args.batch, c, hw, hw)
        x = torch.randn(shape, device=dev)
        dy = torch.randn(shape, device=dev)
        g, b = torch.randn(c, device=dev), torch.randn(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)

        def once():
            o = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
            K.passport_bn_bwd(dy, x, o[1], None, None, 0.0, None, None, None, None, True, True)
        for _ in range(2):
            once()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(args.reps):
            once()
        torch.cuda.synchronize()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
        f = 1000.0 * sum(prof[k][0] for k in prof if k in ('bn_res_fwd', 'bn_stats', 'bn_affine_fwd')) / args.reps
        bw = 1000.0 * sum(prof[k][0] for k in prof if k in ('bn_res_bwd', 'bn_bwd_reduce', 'bn_affine_bwd')) / args.reps
        nf, nb = prof.get('bn_res_fwd', (0, 0))[1] // args.reps, prof.get('bn_res_bwd', (0, 0))[1] // args.reps
        mb = 4.0 * x.numel() / 1e6
        rec = {'C': c, 'HW': hw, 'N': args.batch, 'MB': round(mb, 1), 'count': count, 'fwd_us': round(f, 1), 'bwd_us': round(bw, 1),
               'fwd_launches': nf, 'bwd_launches': nb, 'fwd_TBs': round(2 * mb / f, 2), 'bwd_TBs': round(3 * mb / bw, 2),
               'timeouts': K.sync_timeouts()}
        tot[0] += count * f
        tot[1] += count * bw
        print(json.dumps(rec), flush=True)
        out.append(rec)
    s = {'summary': 'all norm layers of a ResNet50 step (the 7x7 maps of layer4 aside)', 'fwd_ms': round(tot[0] / 1e3, 2),
         'bwd_ms': round(tot[1] / 1e3, 2), 'ranges': os.environ.get('DEEPIPR_BN_RANGES', '1'),
         'stagger': os.environ.get('DEEPIPR_BN_STAGGER', 'default')}
    print(json.dumps(s), flush=True)
    out.append(s)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
