"""Is the vendor convolution run-to-run reproducible on identical inputs, per conv shape of the ResNet18 passport nets?

For every distinct (input shape, weight shape, stride, pad) of ResNet18 at the batch sizes of the bit-identity tests
(64 and 128) this calls aten::convolution and aten::convolution_backward REPS times on the SAME tensors and counts the
distinct results (sha1 of the bytes) of y, dx and dW.  MIOpen is pinned the way the tests pin it
(cudnn.benchmark = False, cudnn.deterministic = True) unless --find.  Between repetitions the caching allocator is
perturbed (--perturb) so that workspaces land on recycled blocks with other contents, which is what a long test session
does and a fresh process does not.

    python tools/conv_determinism.py [--reps 300] [--find] [--perturb] [--prime]
prints one JSON line: {config: {"y": n, "dx": n, "dw": n}} = repetitions whose result differs bit-wise from the first
one, and "nondeterministic": [configs with any]
"""
import argparse
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONVS = [  # (Cin, Cout, H, k, stride, pad)
    (3, 64, 32, 3, 1, 1), (64, 64, 32, 3, 1, 1), (64, 128, 32, 3, 2, 1), (128, 128, 16, 3, 1, 1), (64, 128, 32, 1, 2, 0),
    (128, 256, 16, 3, 2, 1), (256, 256, 8, 3, 1, 1), (128, 256, 16, 1, 2, 0), (256, 512, 8, 3, 2, 1),
    (512, 512, 4, 3, 1, 1), (256, 512, 8, 1, 2, 0),
]


def digest(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=300)
    ap.add_argument('--batches', default='64,128')
    ap.add_argument('--find', action='store_true')
    ap.add_argument('--perturb', action='store_true')
    ap.add_argument('--prime', action='store_true', help='run every conv once in find mode first (fills the find-db the '
                    'way the earlier tests of a session do), then pin')
    args = ap.parse_args()
    dev = 'cuda:0'
    g = torch.Generator(device='cpu').manual_seed(7)
    report, bad = {}, []

    def run(x, w, dy, s, p):
        y = torch.ops.aten.convolution(x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1)
        dx, dw, _ = torch.ops.aten.convolution_backward(dy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1,
                                                        [True, True, False])
        return y, dx, dw

    for n in [int(b) for b in args.batches.split(',')]:
        for ci, co, h, k, s, p in CONVS:
            ho = (h + 2 * p - k) // s + 1
            x = torch.randn(n, ci, h, h, generator=g).to(dev)
            w = (torch.randn(co, ci, k, k, generator=g) * 0.05).to(dev)
            dy = torch.randn(n, co, ho, ho, generator=g).to(dev)
            if args.prime:
                torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = True, False
                run(x, w, dy, s, p)
                torch.cuda.synchronize()
            torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bool(args.find), not args.find
            first, differ = None, {'y': 0, 'dx': 0, 'dw': 0}
            junk = []
            for rep in range(args.reps):
                if args.perturb and rep % 3 == 0:
                    junk = [torch.full((1 << (16 + (rep // 3 + i) % 8),), float(rep + i), device=dev) for i in range(3)]
                    del junk[1]
                    torch.cuda.empty_cache() if rep % 30 == 0 else None
                y, dx, dw = run(x, w, dy, s, p)
                if first is None:
                    first = (y.clone(), dx.clone(), dw.clone())
                    continue
                for key, a, b in zip(('y', 'dx', 'dw'), (y, dx, dw), first):       # every repetition, compared on the GPU
                    if not torch.equal(a, b):
                        differ[key] += 1
            name = 'n%d_%dx%d_%d_k%ds%d' % (n, ci, co, h, k, s)
            report[name] = differ
            if any(differ.values()):
                bad.append(name)
    print(json.dumps({'mode': 'find' if args.find else 'pinned', 'primed': args.prime, 'perturb': args.perturb,
                      'reps': args.reps, 'nondeterministic': bad, 'configs': report}))


if __name__ == '__main__':
    main()
