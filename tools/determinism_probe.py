"""Run-to-run determinism of one ResNet18 passport step on the GPU (MIOpen pinned to immediate mode, as the
bit-identity tests run it): repeats forward + backward REPS times per setting of DEEPIPR_TAIL_FUSION and reports,
per parameter, how many repetitions differ from the first one and by how much.  A gradient that differs only on a
conv weight (and on nothing upstream of it) is the vendor wgrad kernel's summation order; one that differs on
everything upstream of a layer is a norm / passport kernel.

    python tools/determinism_probe.py [--private] [--reps 30]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--private', action='store_true')
    ap.add_argument('--reps', type=int, default=30)
    ap.add_argument('--find', action='store_true', help='leave MIOpen in find mode')
    ap.add_argument('--poison', action='store_true',
                    help='after the first repetition, fill every torch.empty / empty_like buffer of the python layer and '
                         'the kernels\' scratch arena with 0xFF bytes (NaN as float) before use: a kernel that reads '
                         'workspace it did not write shows up as a difference (fresh hipMalloc pages are zero, '
                         'recycled allocator blocks are not)')
    ap.add_argument('--prime', type=int, default=0, help='steps run in MIOpen find mode first, in this process (fills '
                    'its find-db, which immediate mode then consults -- what a long test session does)')
    args = ap.parse_args()
    from tests import test_parity_gpu as T
    if not args.find:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    n, ncls = (64, 100) if args.private else (128, 10)
    prod, _ref, x, y = T._fullsize_pair(args.private, n, ncls)
    x, y = x.to(T.DEV), y.to(T.DEV)
    ce = torch.nn.functional.cross_entropy
    state = {k: v.clone() for k, v in prod.state_dict().items()}
    first = None
    report = {}

    def step():
        prod.load_state_dict(state)
        prod.zero_grad(set_to_none=True)
        if args.private:
            outs = [prod(x, ind=0), prod(x, ind=1)]
            loss = ce(outs[0], y) + ce(outs[1], y)
            loss = loss + sum(m.sign_loss_private.loss for m in prod.modules() if hasattr(m, 'sign_loss_private'))
        else:
            outs = [prod(x)]
            loss = ce(outs[0], y) + sum(m.sign_loss.loss for m in prod.modules()
                                        if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
        loss.backward()
        return outs

    if args.prime:
        saved = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = True, False
        for _ in range(args.prime):
            step()
        torch.cuda.synchronize()
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = saved
    from deepipr_amd import passport_ops
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poisoned(t):
        if t.is_cuda and t.numel() and t.is_contiguous():
            t.view(torch.uint8).fill_(0xFF)
        return t

    for flag in ('1', '0'):
        os.environ['DEEPIPR_TAIL_FUSION'] = flag
        differ, worst, out_differ = {}, {}, 0
        for rep in range(args.reps):
            if args.poison and first is not None:
                torch.empty = lambda *a, **k: poisoned(real_empty(*a, **k))
                torch.empty_like = lambda *a, **k: poisoned(real_empty_like(*a, **k))
                for buf in passport_ops.kernels._arena.values():
                    buf.fill_(0xFF)
            try:
                outs = step()
            finally:
                torch.empty, torch.empty_like = real_empty, real_empty_like
            grads = {k: p.grad.clone() for k, p in prod.named_parameters() if p.grad is not None}
            outs = [o.detach().clone() for o in outs]
            if first is None:
                first = (outs, grads)
                continue
            if any(not torch.equal(a, b) for a, b in zip(outs, first[0])):
                out_differ += 1
            for k, g in grads.items():
                if not torch.equal(g, first[1][k]):
                    differ[k] = differ.get(k, 0) + 1
                    d = float((g - first[1][k]).abs().max())
                    worst[k] = max(worst.get(k, 0.0), d if d == d else float('inf'))
        report[f'tail_fusion={flag}'] = {'reps': args.reps, 'logits_differ': out_differ,
                                         'params': len(first[1]), 'params_differ': differ, 'max_abs_diff': worst}
    passport_ops.kernels.check_exchange()
    print(json.dumps({'private': args.private, 'miopen': 'find' if args.find else 'immediate', 'primed_in_find_mode': args.prime, 'poison': args.poison, **report}))


if __name__ == '__main__':
    main()
