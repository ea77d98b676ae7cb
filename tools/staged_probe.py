"""Where does the staged step's time go?  One rank, RCCL group of one (DEEPIPR_FORCE_DDP=1), config R or P shard.
For each variant: host time to ENQUEUE a step (perf_counter around the loop, no sync) and the step time (with sync).
  graph1     forward + backward as ONE graph, whole exchange after it (round-2 form)
  staged-k   k stages (k graphs), bucket exchanges in between            [--merge collapses stages]
  staged-k-noxchg  the same graphs, every collective / pack skipped until optimizer.step()   (graph-boundary cost alone)
python -m torch.distributed.run --nproc-per-node 1 ... tools/staged_probe.py [--private]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('DEEPIPR_FORCE_DDP', '1')
import bench                                                         # noqa: E402
from deepipr_amd import distributed as D                             # noqa: E402
from deepipr_amd.experiments import staged as S                      # noqa: E402
from deepipr_amd.experiments.graph_step import GraphedTrainStep      # noqa: E402
from deepipr_amd.experiments.trainer import train_step_v1            # noqa: E402
from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23   # noqa: E402
from deepipr_amd.flat_sgd import FlatSGD                             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--private', action='store_true')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--only', default='', help='run one variant only (for a rocprofv3 trace): graph1 | staged-N[-noxchg]')
    a = ap.parse_args()
    rank, local, world = D.init_from_env()
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.backends.cudnn.benchmark = True
    args = argparse.Namespace(arch='resnet18', scheme=2 if a.private else 1, classes=100 if a.private else 10,
                              image_size=32, norm_type='bn', batch=32 if a.private else 128)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(args.batch, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, args.classes, (args.batch,), generator=g).to(dev)
    fn = train_step_v23 if a.private else train_step_v1
    out = {}

    def fresh():
        model = bench.build_model(args, dev)
        model.train()
        with torch.no_grad():
            model(x)
        opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        net = DualBranch(model) if a.private else model
        fn(net, opt, x, y)
        torch.cuda.synchronize()
        return net, opt

    def measure(name, step):
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[name] = {'host_enqueue_ms': round(1000 * (t1 - t0) / a.steps, 3), 'step_ms': round(1000 * (t2 - t0) / a.steps, 3)}

    if a.only in ('', 'graph1'):
        net, opt = fresh()
        g1 = GraphedTrainStep(fn, net, opt, x, y, optimizer_in_graph=False)
        measure('graph1', lambda: g1(x, y))
        del g1, net, opt

    real_plan = S.plan_stages
    for merge in (0, 1, 2):
        def plan(model, optimizer, merge=merge):
            st = real_plan(model, optimizer)
            for _ in range(merge):                     # merge the first two stages (cut of the later one survives)
                if len(st) > 1:
                    st = [S.Stage(st[1].cut, st[0].params + st[1].params)] + st[2:]
            return st
        S.plan_stages = plan
        for noxchg in (False, True):
            name = 'staged-%d%s' % (max(1, 3 - merge), '-noxchg' if noxchg else '')
            if a.only and a.only != name:
                continue
            net, opt = fresh()
            st = S.StagedStep(fn, net, opt, x, y, graph=True)
            if noxchg:
                st._unused_events, st._events = st._events, {}    # the captured event nodes stay (and their events must
                # outlive the graph); nothing is launched behind them
                st._after_stage = lambda k: None
                st._before_stage = lambda k: None
            measure('staged-%d%s' % (len(st.stages), '-noxchg' if noxchg else ''), lambda: st(x, y))
            del st, net, opt
    S.plan_stages = real_plan
    print(json.dumps(out))
    D.shutdown()


if __name__ == '__main__':
    main()
