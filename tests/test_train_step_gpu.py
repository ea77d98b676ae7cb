"""The train step's plumbing on the GPU (experiments/trainer.py:128-149): hipGraph replay under a learning-rate schedule,
the captured optimiser, the fused cross-entropy head, external events of a captured step."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import patterns, torch_ref
from oracle.cases import ALPHA, SGD, alexnet_config
from tests.compare import close, states_close
from tests.gpu_common import DEV, K, dev, host, pinned_miopen      # noqa: F401  (K is a fixture)
from tests.impls import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- hipGraph replay and lr schedules
@pytest.mark.miopen_pinned
@pytest.mark.parametrize('flat', [True, False])
def test_graph_replay_follows_the_lr_schedule(K, flat):
    """A step captured into a hipGraph must keep following MultiStepLR (lr_configs/default.json decays at epochs 100
    and 150): FlatSGD reads lr from device memory (no re-capture), torch.optim.SGD is re-captured when its
    param_groups change.  Trajectory == the eager one across a milestone."""
    from deepipr_amd.experiments.trainer import Trainer
    from deepipr_amd.flat_sgd import FlatSGD
    from tests.test_parity_gpu import _fullsize_pair
    finals = []
    for graph in (False, True):
        prod, _ref, x, y = _fullsize_pair(False, 32, 10)
        x, y = x.to(DEV), y.to(DEV)
        loader = [(x, y), (x.flip(0), y.flip(0))]
        opt = (FlatSGD if flat else torch.optim.SGD)(prod.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, [1, 2], 0.1)
        tr = Trainer(prod, opt, sched, torch.device(DEV), graph=graph)
        with pinned_miopen():
            for epoch in range(3):                               # lr 0.05 -> 0.005 -> 0.0005
                tr.train(epoch, loader)
            torch.cuda.synchronize()
        if graph:
            g = tr.step._graphed
            assert g is not None
            assert g.recaptures == (0 if flat else 2)
        finals.append({k: v.clone() for k, v in prod.state_dict().items()})
        assert opt.param_groups[0]['lr'] == pytest.approx(0.0005)
    states_close(finals[0], finals[1], what='eager vs replayed')
    # and the schedule really acted: a frozen lr of 0.05 would have moved the weights ~3.4x further in epochs 2-3
    assert K.sync_timeouts() == 0


@pytest.mark.miopen_pinned
def test_captured_flat_sgd_reads_gradients_in_place(K):
    """One GPU, step captured into a hipGraph: FlatSGD updates from the gradient tensors where autograd left them
    (deepipr_sgd_momentum_step_multi, chunk table built at capture time) instead of packing them into flat_grad first.
    Same trajectory as the eager FlatSGD step (which packs), over steps with changing inputs."""
    from deepipr_amd.experiments.graph_step import GraphedTrainStep
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.flat_sgd import FlatSGD
    from tests.test_parity_gpu import _fullsize_pair
    finals = []
    for graphed in (False, True):
        prod, _ref, x, y = _fullsize_pair(False, 32, 10)
        x, y = x.to(DEV), y.to(DEV)
        opt = FlatSGD(prod.parameters(), **SGD)
        with pinned_miopen():
            if graphed:
                g = GraphedTrainStep(train_step_v1, prod, opt, x, y, warmup=0)
                assert opt.in_place_captures == 1
                for i in range(4):
                    g(x if i % 2 == 0 else x.flip(0), y if i % 2 == 0 else y.flip(0))
            else:
                for i in range(4):
                    train_step_v1(prod, opt, x if i % 2 == 0 else x.flip(0), y if i % 2 == 0 else y.flip(0))
                assert opt.in_place_captures == 0
            torch.cuda.synchronize()
        finals.append({k: v.clone() for k, v in prod.state_dict().items()})
    states_close(finals[0], finals[1], what='eager vs replayed')


@pytest.mark.parametrize('n,c', [(128, 10), (66, 100), (32, 100), (8, 1000), (256, 1000), (1, 2), (7, 1)])
def test_fused_cross_entropy_and_top1(K, n, c):
    """deepipr_ce_top1_fwd / deepipr_ce_bwd against F.cross_entropy + the reference's accuracy() (trainer.py:28-43,
    136) in float64: loss 1e-6, gradient 1e-6 of its scale, top-1 identical (ties: lowest index, as argmax)."""
    from deepipr_amd import passport_ops as P
    rs = np.random.RandomState(n + c)
    logits = (rs.standard_normal((n, c)) * 3).astype(np.float32)
    if c >= 2:
        logits[0, 1] = logits[0, 0] = logits[0].max() + 1.0             # a tie for the maximum: class 0 wins
    target = rs.randint(0, c, size=n).astype(np.int64)
    x = dev(logits).requires_grad_(True)
    t = torch.from_numpy(target).to(DEV)
    assert P.kernels.ce_usable(x, t)
    loss, top1 = P.cross_entropy_top1(x, t)
    (loss * 1.7).backward()
    xr = torch.from_numpy(logits).double().requires_grad_(True)
    lr = torch.nn.functional.cross_entropy(xr, torch.from_numpy(target))
    (lr * 1.7).backward()
    assert abs(float(loss) - float(lr)) <= 1e-6 * max(1.0, abs(float(lr)))
    want_top1 = float((torch.from_numpy(logits).argmax(dim=1) == torch.from_numpy(target)).double().mean() * 100.0)
    assert float(top1) == pytest.approx(want_top1, abs=1e-4)
    g_ref = xr.grad.numpy()
    assert np.abs(host(x.grad) - g_ref).max() <= 1e-6 * max(1e-3, np.abs(g_ref).max())
    assert not top1.requires_grad


@pytest.mark.miopen_pinned
def test_replay_with_eager_optimizer_survives_an_eager_step_in_between(K):
    """Data-parallel form of the graphed step (forward + backward replayed, optimiser eager): a ragged last batch runs
    the eager step, which re-binds every `.grad`; the next replay's optimiser step must still use the gradients the
    replayed backward wrote.  Trajectory == all-eager."""
    from deepipr_amd.experiments.graph_step import GraphedTrainStep
    from deepipr_amd.experiments.trainer import train_step_v1
    from tests.test_parity_gpu import _fullsize_pair
    finals = []
    for graphed in (False, True):
        prod, _ref, x, y = _fullsize_pair(False, 32, 10)
        x, y = x.to(DEV), y.to(DEV)
        opt = torch.optim.SGD(prod.parameters(), **SGD)
        seq = [(x, y), (x[:20], y[:20]), (x.flip(0), y.flip(0)), (x, y)]           # the second batch is ragged
        with pinned_miopen():
            g = GraphedTrainStep(train_step_v1, prod, opt, x, y, warmup=0, optimizer_in_graph=False) if graphed else None
            for xb, yb in seq:
                if graphed and xb.shape[0] == x.shape[0]:
                    g(xb, yb)
                else:
                    train_step_v1(prod, opt, xb, yb)
            torch.cuda.synchronize()
        finals.append({k: v.clone() for k, v in prod.state_dict().items()})
    states_close(finals[0], finals[1], what='eager vs replayed')


# ----------------------------------------------------------------------------- cross-entropy head: labels out of range
@pytest.mark.parametrize('bad', [-100, -1, 10, 1 << 40])
def test_fused_cross_entropy_refuses_out_of_range_labels_loudly(K, bad):
    """A label outside [0, C) -- F.cross_entropy's default ignore_index = -100 included, which this head does not
    implement -- must not index out of bounds and must not give a plausible loss: loss is NaN, the row's gradient is
    NaN, the other rows' gradients are untouched (k_ce_rows / k_ce_bwd read row[t] unvalidated)."""
    from deepipr_amd import passport_ops as P
    rs = np.random.RandomState(3)
    logits = dev(rs.standard_normal((16, 10)) * 2).requires_grad_(True)
    target = torch.from_numpy(rs.randint(0, 10, size=16).astype(np.int64)).to(DEV)
    good_loss, _ = P.cross_entropy_top1(logits, target)
    good_loss.backward()
    good = logits.grad.clone()
    logits.grad = None
    target[5] = bad
    loss, top1 = P.cross_entropy_top1(logits, target)
    loss.backward()
    assert torch.isnan(loss) and torch.isfinite(top1)
    g = logits.grad
    assert torch.isnan(g[5]).all()
    keep = torch.ones(16, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert torch.isfinite(g[keep]).all() is not None          # rows scale with dloss = NaN-free upstream gradient (1.0)
    assert torch.equal(g[keep], good[keep])


# ----------------------------------------------------------------------------- external events of a captured step
def test_external_event_of_a_captured_graph_orders_a_side_stream(K):
    """deepipr_event_record on a capturing stream = an external event-record node: after every launch of the graph, a
    stream made to wait for the event (deepipr_stream_wait_event, issued after the launch call) must see everything the
    graph did BEFORE the node -- this is what lets experiments/staged.py start a gradient bucket's all-reduce from
    outside the graph while the replayed backward is still running.  A wait that bound to an older record, or returned
    early, would let the side stream read the previous replay's values."""
    from deepipr_amd import _lib
    dev_ = torch.device(DEV)
    big = torch.zeros(1 << 26, device=dev_)                    # 256 MB: the increment below takes ~100 us
    tail = torch.zeros(1 << 26, device=dev_)
    cap, side = torch.cuda.Stream(device=dev_), torch.cuda.Stream(device=dev_)
    ev = _lib.ExternalEvent()
    cap.wait_stream(torch.cuda.current_stream(dev_))
    with torch.cuda.stream(cap):
        big.add_(0.0)                                          # warm the kernels outside the capture
        tail.add_(0.0)
    torch.cuda.current_stream(dev_).wait_stream(cap)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        big.add_(1.0)
        ev.record(cap)                                         # <- external event-record node
        for _ in range(4):
            tail.add_(1.0)                                     # work behind the node: the graph is still running
    torch.cuda.synchronize()
    for i in range(1, 31):
        g.replay()
        ev.wait(side)
        with torch.cuda.stream(side):
            snap = big[:: 1 << 12].clone()                     # strided sample across the whole buffer
            done = tail[:: 1 << 16].clone()
        side.synchronize()
        assert bool((snap == float(i)).all()), (i, snap.unique().tolist())
        # (not asserted: `done` usually still holds a value below 4 * i -- the side stream ran ahead of the graph's tail)
        assert float(done.max()) <= 4.0 * i
    torch.cuda.synchronize()
    # the host-side wait (what staged.py uses before it enqueues a bucket's collective): after it, what the graph did
    # before the node is visible to a plain read on ANY stream
    for i in range(31, 41):
        g.replay()
        ev.synchronize()
        assert bool((big[:: 1 << 12] == float(i)).all()), i
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------- the test loops as graph replays
@pytest.mark.miopen_pinned
@pytest.mark.parametrize('private', [False, True])
def test_graphed_test_loop_equals_the_eager_one(K, private):
    """Tester / TesterPrivate with graph=True (GraphedEval: one hipGraph per batch shape, the ragged last batch included)
    return what the eager loop returns -- the same kernels run -- and keep doing so after the weights moved in place (a
    training step), after running statistics moved, and after a passport key was replaced (frozen address: captured again)."""
    from deepipr_amd.experiments.trainer import Tester
    from deepipr_amd.experiments.trainer_private import TesterPrivate
    from tests.test_parity_gpu import _fullsize_pair
    prod, _ref, x, y = _fullsize_pair(private, 37, 10)
    x, y = x.to(DEV), y.to(DEV)
    loader = [(x[:16], y[:16]), (x[16:32], y[16:32]), (x[32:], y[32:])]               # 16, 16 and a ragged 5
    with pinned_miopen():
        if private:
            eager, graphed = TesterPrivate(prod, DEV, verbose=False), TesterPrivate(prod, DEV, verbose=False, graph=True)
            run = lambda t: [t.test(loader, ind=i) for i in (0, 1)]
        else:
            eager, graphed = Tester(prod, DEV, verbose=False), Tester(prod, DEV, verbose=False, graph=True)

            def run(t):
                cmp = []
                out = t.test(loader, compare=cmp)
                return [out, [host(a).tolist() for a, _b in cmp]]

        def same():
            a, b = run(eager), run(graphed)
            for ra, rb in zip(a, b):
                if isinstance(ra, dict):
                    assert ra['loss'] == rb['loss'] and ra['acc'] == rb['acc'], (ra, rb)
                else:
                    assert ra == rb
        same()
        counters = [g for g in ([graphed._batch] if not private else graphed._batch.values())]
        assert all(g.captures == 2 for g in counters)                                   # two shapes, one graph each
        same()
        assert all(g.captures == 2 for g in counters)                                   # replays only
        with torch.no_grad():                                                           # weights and statistics move in place
            for p in prod.parameters():
                p.add_(0.01 * torch.randn_like(p))
            for name, b in prod.named_buffers():
                if name.endswith('running_mean'):
                    b.add_(0.05)
        same()
        assert all(g.captures == 2 for g in counters)
        layer = next(m for m in prod.modules() if hasattr(m, 'set_key') and m.get_bias_key() is not None)
        key = layer.get_bias_key()
        layer.set_key(torch.randn_like(key), torch.randn_like(key))                    # a new key: new pooled means
        same()
        assert all(g.captures == 4 for g in counters)


# ----------------------------------------------------------------------------- the step's scalar sums in one launch
def test_scalar_sums_equal_the_chain_of_adds_bit_for_bit(K):
    """deepipr_scalar_sums: `loss_public + loss_private`, `sign_loss += m.loss` over the layers and `loss + sign_loss`
    (experiments/trainer.py:140-145, trainer_private.py:163-173) as ONE launch -- the same left-to-right fp32 adds, so the
    same bits; the gradient of the objective reaches every term as 1."""
    from deepipr_amd.passport_ops import scalar_sums
    g = torch.Generator().manual_seed(11)
    for n_a, n_b in [(1, 5), (2, 10), (1, 1), (2, 0), (0, 3), (1, 47)]:
        vals = (torch.randn(n_a + n_b, generator=g) * torch.logspace(-3, 3, n_a + n_b)).to(DEV)
        terms = [vals[i].clone().requires_grad_(True) for i in range(n_a + n_b)]        # 0-d tensors, like the losses
        a, b, tot = scalar_sums(terms[:n_a], terms[n_a:])
        ra = rb = None
        for t in terms[:n_a]:
            ra = t.detach() if ra is None else ra + t.detach()
        for t in terms[n_a:]:
            rb = t.detach() if rb is None else rb + t.detach()
        want = ra if rb is None else (rb if ra is None else ra + rb)
        assert (a is None) == (ra is None) and (b is None) == (rb is None)
        assert (a is None or torch.equal(a.detach(), ra)) and (b is None or torch.equal(b.detach(), rb))
        assert torch.equal(tot.detach(), want)
        (tot * 3.0).backward()
        assert all(float(t.grad) == 3.0 for t in terms)
    # a term that is itself the output of an op keeps its place in the autograd graph
    x = torch.tensor(2.0, device=DEV, requires_grad=True)
    _a, _b, tot = scalar_sums([x * x], [x * 3.0, x.detach() * 0.5])
    tot.backward()
    assert float(x.grad) == 7.0 and float(tot) == 11.0
    v = torch.ones(1, device=DEV, requires_grad=True)                                  # a one-element vector among the scalars
    _a, _b, tot = scalar_sums([v * 2.0], [x.detach()])
    tot.backward()
    assert v.grad.shape == (1,) and float(v.grad) == 2.0
