"""Passport GEMV pair on the five layer4 weights of ResNet18 (33.55 MB): per-layer launches vs ONE batched launch,
forward (gamma / beta: W read once) and backward (rank-2 update accumulated into dW: read + write), timed with the
library's per-dispatch events.  python tools/gemv_bench.py [--reps 50]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepipr_amd import _lib                                # noqa: E402
from deepipr_amd.passport_ops import kernels as K           # noqa: E402

SHAPES = [(512, 256, 3), (512, 512, 3), (512, 256, 1), (512, 512, 3), (512, 512, 3)]      # layer4: (Co, Ci, k)


def timed(fn, reps, key):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms, n = _lib.profile_read()[key]
    nbytes = _lib.profile_read_bytes()[key]
    _lib.profile_enable(False)
    return {'launches_per_rep': n / reps, 'us_per_rep': round(1000.0 * ms / reps, 2), 'avg_us': round(1000.0 * ms / n, 2),
            'bytes_per_rep': int(nbytes / reps), 'GBps': round(nbytes / (ms * 1e-3) / 1e9, 1),
            'frac_of_8TBps': round(nbytes / (ms * 1e-3) / 8e12, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--flush', action='store_true', help='stream 1 GB through the caches between repetitions')
    args = ap.parse_args()
    dev = 'cuda:0'
    ws = [torch.randn(co, ci, k, k, device=dev) * 0.02 for co, ci, k in SHAPES]
    ms = [torch.randn(2, w[0].numel(), device=dev, dtype=torch.float64) for w in ws]
    dgs = [torch.randn(w.shape[0], device=dev) for w in ws]
    dbs = [torch.randn(w.shape[0], device=dev) for w in ws]
    dws = [torch.randn_like(w) for w in ws]
    junk = torch.empty(1 << 28, device=dev) if args.flush else None

    def flush():
        if junk is not None:
            junk.add_(1.0)
    out = {}
    if _lib.has_test_hooks():                           # DEEPIPR_LIB = the test build: sweep the rows-per-workgroup forms
        for rows in (1, 2):
            os.environ['DEEPIPR_GEMV_ROWS'] = str(rows)
            out['fwd_batched_rows%d' % rows] = timed(lambda: (flush(), K.gamma_beta_fwd_multi(ws, ms)), args.reps, 'gamma_beta_fwd')
            out['fwd_per_layer_rows%d' % rows] = timed(lambda: (flush(), [K.gamma_beta_fwd(w, m) for w, m in zip(ws, ms)]), args.reps, 'gamma_beta_fwd')
        del os.environ['DEEPIPR_GEMV_ROWS']
    out['fwd_per_layer'] = timed(lambda: (flush(), [K.gamma_beta_fwd(w, m) for w, m in zip(ws, ms)]), args.reps, 'gamma_beta_fwd')
    out['fwd_batched'] = timed(lambda: (flush(), K.gamma_beta_fwd_multi(ws, ms)), args.reps, 'gamma_beta_fwd')
    out['bwd_acc_per_layer'] = timed(lambda: (flush(), [K.gamma_beta_bwd_acc(g, b, m, d) for g, b, m, d in zip(dgs, dbs, ms, dws)]),
                                     args.reps, 'gamma_beta_bwd')
    out['bwd_acc_batched'] = timed(lambda: (flush(), K.gamma_beta_bwd_multi(dgs, dbs, ms, dws, True)), args.reps, 'gamma_beta_bwd')
    # same values from both forms
    a = [K.gamma_beta_fwd(w, m) for w, m in zip(ws, ms)]
    b = K.gamma_beta_fwd_multi(ws, ms)
    out['batched_equals_per_layer'] = all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, b))
    print(json.dumps(out))


if __name__ == '__main__':
    main()
