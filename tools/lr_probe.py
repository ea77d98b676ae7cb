"""Eager vs replayed trajectory of the ResNet18 V1 step (FlatSGD, lr 0.05 -> the setting of
tests/test_train_step_gpu.py::test_graph_replay_follows_the_lr_schedule, which was seen to fail once right after the
whole-net parity tests): after EVERY step the two models' parameters are compared bit by bit.  The two paths run the same
kernels on the same inputs, so they must stay bit-identical; the first step at which they do not, and the tensors that
differ first, localise a nondeterminism.   python tools/lr_probe.py [--reps 10] [--prelude]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--prelude', action='store_true', help='run the whole-net parity tests in this process first')
    args = ap.parse_args()
    if args.prelude:
        import pytest
        pytest.main(['-q', '-x', '-p', 'no:cacheprovider', os.path.join(ROOT, 'tests', 'test_models_gpu.py'), '-k', 'whole_net'])
    from deepipr_amd.experiments.graph_step import GraphedTrainStep
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.flat_sgd import FlatSGD
    from tests import test_parity_gpu as T
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    out = []
    for rep in range(args.reps):
        a, _r, x, y = T._fullsize_pair(False, 32, 10)
        b, _r, _x, _y = T._fullsize_pair(False, 32, 10)
        x, y = x.to(T.DEV), y.to(T.DEV)
        oa = FlatSGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        ob = FlatSGD(b.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        g = GraphedTrainStep(train_step_v1, b, ob, x, y, warmup=0)
        rec = {'rep': rep, 'first_diff_step': None}
        for step in range(args.steps):
            xs, ys = (x, y) if step % 2 == 0 else (x.flip(0), y.flip(0))
            la = train_step_v1(a, oa, xs, ys)
            lb = g(xs, ys)
            torch.cuda.synchronize()
            diff = {k: float((p - q).abs().max()) for (k, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items())
                    if p.dtype.is_floating_point and not torch.equal(p, q)}
            if diff:
                rec['first_diff_step'] = step
                rec['n_differ'] = len(diff)
                rec['loss_equal'] = bool(torch.equal(la[0], lb[0]))
                rec['largest'] = sorted(diff.items(), key=lambda kv: -kv[1])[:6]
                rec['smallest_named'] = sorted(diff)[:6]
                break
        out.append(rec)
        del g, a, b, oa, ob
    print(json.dumps({'prelude': args.prelude, 'reps': args.reps, 'mismatching_reps': sum(r['first_diff_step'] is not None for r in out),
                      'records': out}))


if __name__ == '__main__':
    main()
