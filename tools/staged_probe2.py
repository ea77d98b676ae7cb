"""Which ingredient of the staged capture costs time at replay?  One rank, RCCL group of one, config R; every variant
measured twice, interleaved (clock drift shows as a difference between the two passes).
  bwd        GraphedTrainStep: loss.backward() in one graph, exchange after it            (the round-2 form)
  grad       the same, but the captured step calls torch.autograd.grad and assigns p.grad
  staged1    StagedStep with ONE stage (no cut, no event)
  staged1s   the same with FlatSGD put back into its 'single' exchange mode after the capture
  staged3    StagedStep, three stages in one graph, two external events, bucket exchanges on the side stream
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEEPIPR_FORCE_DDP', '1')
import argparse                                                       # noqa: E402
import bench                                                          # noqa: E402
from deepipr_amd import distributed as D                              # noqa: E402
from deepipr_amd.experiments import staged as S                       # noqa: E402
from deepipr_amd.experiments.graph_step import GraphedTrainStep       # noqa: E402
from deepipr_amd.experiments.trainer import forward_loss_v1, train_step_v1   # noqa: E402
from deepipr_amd.flat_sgd import FlatSGD                              # noqa: E402


def step_with_autograd_grad(model, optimizer, data, target):
    optimizer.zero_grad(set_to_none=True)
    objective, out = forward_loss_v1(model, data, target)
    params = [p for g in optimizer._opt.param_groups for p in g['params']] if hasattr(optimizer, '_opt') else \
        [p for g in optimizer.param_groups for p in g['params']]
    grads = torch.autograd.grad([objective], params, allow_unused=True)
    for p, g in zip(params, grads):
        p.grad = g
    optimizer.step()
    return out


def main():
    rank, local, world = D.init_from_env()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    private = '--private' in sys.argv
    args = argparse.Namespace(arch='resnet18', scheme=2 if private else 1, classes=100 if private else 10, image_size=32, norm_type='bn', batch=32 if private else 128)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, args.classes, (args.batch,), generator=g).to(dev)

    def fresh():
        model = bench.build_model(args, dev)
        model.train()
        with torch.no_grad():
            model(x)
        opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        train_step_v1(model, opt, x, y)
        torch.cuda.synchronize()
        return model, opt

    real_plan = S.plan_stages
    one = lambda model, optimizer: [S.Stage(None, [p for g in optimizer.param_groups for p in g['params']])]
    steps = {}
    m, o = fresh(); g1 = GraphedTrainStep(train_step_v1, m, o, x, y, optimizer_in_graph=False); steps['bwd'] = g1
    m, o = fresh(); g2 = GraphedTrainStep(step_with_autograd_grad, m, o, x, y, optimizer_in_graph=False); steps['grad'] = g2
    S.plan_stages = one
    m, o = fresh(); s1 = S.StagedStep(train_step_v1, m, o, x, y, graph=True); steps['staged1'] = s1
    m, o = fresh(); s1s = S.StagedStep(train_step_v1, m, o, x, y, graph=True); o._mode = 'single'; steps['staged1s'] = s1s
    S.plan_stages = real_plan
    m, o = fresh(); s3 = S.StagedStep(train_step_v1, m, o, x, y, graph=True); steps['staged3'] = s3
    # the same graph (two event-record nodes inside), nothing launched behind the events: everything goes out in step()
    m, o = fresh(); s3n = S.StagedStep(train_step_v1, m, o, x, y, graph=True); s3n._kept, s3n._events = s3n._events, {}
    steps['staged3_events_unused'] = s3n
    # buckets exchanged behind their events, but without the collective (pack only) / without the pack (collective only)
    m, o = fresh(); s3p = S.StagedStep(train_step_v1, m, o, x, y, graph=True); o.comm = False; steps['staged3_pack_only'] = s3p
    out = {k: [] for k in steps}
    for rnd in range(3):
        for name, st in steps.items():
            for _ in range(10):
                st(x, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(60):
                st(x, y)
            torch.cuda.synchronize()
            out[name].append(round(1000 * (time.perf_counter() - t0) / 60, 3))
    print(json.dumps(out))
    D.shutdown()


if __name__ == '__main__':
    main()
