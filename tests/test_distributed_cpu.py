"""N>1 path on CPU: world_size-2 `gloo` processes drive the product's replicate() / trainers.

The HIP kernels cannot run here, so each worker monkeypatches passport_ops.kernels with the oracle-backed
stand-in (tests/oracle_kernels.py); everything else -- key materialisation check, rank-0 state broadcast,
DistributedDataParallel wrapping, the V1 and V2 (DualBranch) steps -- is the shipped code.

Checked: (1) replicas start identical to rank 0 (weights, keys, signature bits) although every rank was
built with a different seed; (2) one data-parallel step on two half-batches equals one single-process step
on the full batch (norm_type='none', so no per-shard batch statistics enter); (3) ranks stay in lock step.
"""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(private, seed):
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.alexnet_passport import AlexNetPassport
    from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
    from oracle.cases import alexnet_config
    kw = construct_passport_kwargs_from_dict({'passport_config': alexnet_config(), 'norm_type': 'none',
                                              'key_type': 'random', 'sl_ratio': 0.1})
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = (AlexNetPassportPrivate if private else AlexNetPassport)(3, 10, kw)
    model.train()
    with torch.no_grad():
        model(torch.randn(2, 3, 32, 32))          # materialise the random keys (rank-dependent on purpose)
    return model


def _batch():
    g = torch.Generator().manual_seed(77)
    return torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 10, (8,), generator=g)


def _worker(rank, world, port, private, out_dir, flat=False):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from deepipr_amd import distributed as D
    from deepipr_amd import passport_ops
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23
    from tests.oracle_kernels import OracleKernels
    passport_ops.kernels = OracleKernels()
    r, lr, w = D.init_from_env('gloo')
    assert (r, w) == (rank, world)
    model = _build(private, seed=100 + rank)       # different weights / keys / bits per rank before the sync
    dev = torch.device('cpu')
    if flat:                                       # FlatSGD: no DDP wrapper, the optimiser exchanges the gradients
        from deepipr_amd.flat_sgd import FlatSGD
        D.check_keys_materialised(model)
        D.broadcast_state(model, 0)
        wrapped = DualBranch(model) if private else model
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        opt = FlatSGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
        assert len(opt._buckets) >= 3 and opt.world == 2
    else:
        wrapped = D.replicate(DualBranch(model) if private else model, dev)
        state0 = {k: v.clone() for k, v in model.state_dict().items()}
        opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    # with a gradient exchange launched from inside backward, the co-residency-dependent single-pass kernels are withheld
    assert passport_ops.kernels.allow_sync is False
    x, y = _batch()
    lo, hi = rank * 4, rank * 4 + 4
    step = train_step_v23 if private else train_step_v1
    if flat == 'replay':
        # what a replayed hipGraph of forward + backward leaves behind (experiments/graph_step.py, data-parallel
        # mode): gradients, no hook ever fired, the optimiser runs eagerly afterwards -- one all-reduce per step
        from deepipr_amd.experiments.graph_step import _NoStep
        inner, calls, real_all_reduce = step, [], dist.all_reduce

        def counted(t, *a, **k):
            calls.append(t.numel())
            return real_all_reduce(t, *a, **k)

        opt.set_mode('single')                    # what GraphedTrainStep._capture does: rank-invariant, no hooks
        assert passport_ops.kernels.allow_sync is True      # ... and nothing overlaps the captured kernels any more

        def step(wrapped, opt, xs, ys):
            out = inner(wrapped, _NoStep(opt), xs, ys)
            dist.all_reduce = counted
            try:
                opt.step()
            finally:
                dist.all_reduce = real_all_reduce
            assert calls == [opt.flat_grad.numel()], calls
            calls.clear()
            return out
    if flat == 'staged':
        # experiments/staged.py, eager form: backward stage by stage, stage k's bucket all-reduced while stage k + 1
        # back-propagates, the last one in optimizer.step() -- the collectives' order and sizes follow the stage plan
        from deepipr_amd.experiments.staged import StagedStep
        calls, real_all_reduce = [], dist.all_reduce

        def counted(t, *a, **k):
            calls.append(t.numel())
            return real_all_reduce(t, *a, **k)
        staged = StagedStep(step, wrapped, opt, x[lo:hi], y[lo:hi], graph=False, warmup=0)
        assert opt._mode == 'staged' and len(staged.stages) == 2 and [s.cut for s in staged.stages] == ['features.5', None]
        assert not [k for k, v in model.state_dict().items() if not torch.equal(v, state0[k])]   # warmup=0: a dry run

        def step(wrapped, opt, xs, ys):
            dist.all_reduce = counted
            try:
                out = staged(xs, ys)
            finally:
                dist.all_reduce = real_all_reduce
            assert len(calls) == 2 and sum(calls) == opt.flat_grad.numel(), calls
            calls.clear()
            return out
    out = step(wrapped, opt, x[lo:hi], y[lo:hi])
    if flat:                                       # a second step exercises momentum and the bucket bookkeeping
        out2 = step(wrapped, opt, x[lo:hi].flip(0), y[lo:hi].flip(0))
    torch.save({'state0': state0, 'state1': {k: v.clone() for k, v in model.state_dict().items()},
                'sign_loss': float(out[1])}, os.path.join(out_dir, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('flat', [False, True, 'replay', 'staged'])
@pytest.mark.parametrize('private', [False, True])
def test_two_rank_step_equals_single_process_step(private, flat, tmp_path, monkeypatch):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, private, str(tmp_path), flat), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    # (1) after replicate(): every rank holds rank 0's weights, keys and signature bits
    for k in r0['state0']:
        assert torch.equal(r0['state0'][k], r1['state0'][k]), k
    assert any(k.endswith('key') or k.endswith('key_private') for k in r0['state0'])
    # (3) ranks stay identical after the step
    for k in r0['state1']:
        assert torch.equal(r0['state1'][k], r1['state1'][k]), k
    assert r0['sign_loss'] == pytest.approx(r1['sign_loss'], rel=1e-6)

    # (2) the same step in one process on the full batch, starting from rank 0's synchronised state
    from deepipr_amd import passport_ops
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23
    from tests.oracle_kernels import OracleKernels
    monkeypatch.setattr(passport_ops, 'kernels', OracleKernels())
    torch.set_num_threads(4)
    model = _build(private, seed=5)
    model.load_state_dict(r0['state0'])
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    x, y = _batch()
    if private:
        out = train_step_v23(DualBranch(model), opt, x, y)
    else:
        out = train_step_v1(model, opt, x, y)
    assert float(out[1]) == pytest.approx(r0['sign_loss'], rel=1e-5)
    if flat:
        xf = torch.cat([x[:4].flip(0), x[4:].flip(0)])
        yf = torch.cat([y[:4].flip(0), y[4:].flip(0)])
        (train_step_v23(DualBranch(model), opt, xf, yf) if private else train_step_v1(model, opt, xf, yf))
    single = model.state_dict()
    for k, v in r0['state1'].items():
        if v.dtype.is_floating_point:
            assert torch.allclose(single[k], v, rtol=1e-4, atol=1e-6), (k, float((single[k] - v).abs().max()))


def test_replicate_refuses_unset_keys():
    from deepipr_amd import distributed as D
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.alexnet_passport import AlexNetPassport
    from oracle.cases import alexnet_config
    kw = construct_passport_kwargs_from_dict({'passport_config': alexnet_config(), 'norm_type': 'bn',
                                              'key_type': 'random', 'sl_ratio': 0.1})
    with pytest.raises(RuntimeError, match='passport keys'):
        D.check_keys_materialised(AlexNetPassport(3, 10, kw))


def _entry_worker(rank, world, port, private, ddp, logdir):
    sys.path.insert(0, ROOT)
    os.chdir(ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from deepipr_amd import passport_ops
    from tests.oracle_kernels import OracleKernels
    passport_ops.kernels = OracleKernels()
    import train_v1
    import train_v23
    argv = ['--arch', 'alexnet', '--key-type', 'random', '--epochs', '2', '--batch-size', '8',
            '--synthetic-samples', '16', '--device', 'cpu', '--backend', 'gloo', '--logdir', logdir, '--norm-type', 'none']
    argv += ['--train-backdoor'] if private else ['--train-passport']
    argv += ['--ddp'] if ddp else []
    out = (train_v23 if private else train_v1).main(argv)
    torch.save({'logdir': out['logdir'], 'rows': len(out['history']),
                'state': {k: v.clone() for k, v in out['model'].state_dict().items()}},
               os.path.join(logdir, 'entry_rank%d.pt' % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize('private,ddp', [(False, False), (True, False), (False, True)])
def test_entry_points_with_two_ranks(private, ddp, tmp_path):
    """train_v1.py / train_v23.py --train-backdoor as two gloo ranks (the flow torchrun starts on a multi-GPU node):
    process-group set-up from the environment, rank-0 state broadcast, sharded synthetic loaders, FlatSGD's (or DDP's)
    gradient exchange, rank-0-only evaluation and checkpoints between barriers, the experiment id agreed by broadcast."""
    port = _free_port()
    mp.spawn(_entry_worker, args=(2, port, private, ddp, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / 'entry_rank0.pt')
    r1 = torch.load(tmp_path / 'entry_rank1.pt')
    assert r0['logdir'] == r1['logdir'] and r0['logdir'].endswith(os.sep + '1')
    assert r0['rows'] == 2 and r1['rows'] == 0                  # history and checkpoints are rank 0's
    assert os.path.exists(os.path.join(r0['logdir'], 'history.csv'))
    # the gradient exchange kept the replicas in lock step over both epochs (norm_type 'none': no per-rank statistics),
    # and what rank 0 saved is that state
    saved = torch.load(os.path.join(r0['logdir'], 'models', 'last.pth'))
    assert any(k.endswith('key') or k.endswith('key_private') for k in saved)
    for k, v in r0['state'].items():
        assert torch.equal(v, r1['state'][k]), k
        assert torch.equal(v, saved[k]), k


def _find_phase_worker(rank, world, port, out_dir):
    """rank0_first / gradients_agree / ranks_seen, the helpers bench.py's find phase is made of (two gloo ranks)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import time
    from deepipr_amd import distributed as D
    D.init_from_env('gloo')
    dev = torch.device('cpu')
    stamp = os.path.join(out_dir, 'rank0_done')

    def first_convolutions():
        if rank == 0:
            time.sleep(0.5)                                  # rank 0's "find" takes a while ...
            open(stamp, 'w').write('db')
            return 'measured'
        return 'from the database' if os.path.exists(stamp) else 'raced'      # ... the others must come after it
    res, seconds = D.rank0_first(first_convolutions, dev)
    seen = D.ranks_seen(dev)
    same = [torch.arange(6.0).view(2, 3), None, torch.ones(4)]
    agree_same = D.gradients_agree(same, dev)
    diff = [torch.arange(6.0).view(2, 3) + (1e-3 if rank else 0.0), None, torch.ones(4)]
    agree_diff = D.gradients_agree(diff, dev)
    near = [torch.arange(1.0, 7.0).view(2, 3) * (1.0 + (1e-7 if rank else 0.0)), torch.ones(4)]
    agree_near = D.gradients_agree(near, dev)
    # a norm layer's shift gradient (1-D, a sum that cancels): reported in LAST_AGREEMENT, not judged
    shift = [torch.arange(6.0).view(2, 3), torch.tensor([1.0, -1.0 + (1e-3 if rank else 0.0)])]
    agree_shift = D.gradients_agree(shift, dev)
    detail_shift = dict(D.LAST_AGREEMENT)
    torch.save(dict(res=res, seconds=seconds, seen=seen, agree_same=agree_same, agree_diff=agree_diff,
                    agree_near=agree_near, agree_shift=agree_shift, detail_shift=detail_shift),
               os.path.join(out_dir, 'r%d.pt' % rank))
    D.shutdown()


def test_find_phase_helpers_rank0_first_then_the_others(tmp_path):
    """bench.py's find phase for N > 1: rank 0 runs the first convolutions alone, the others behind a
    barrier (they then read rank 0's MIOpen find records instead of racing on the user database); a SUM all-reduce of ones
    counts the ranks a collective really reaches; the ranks' gradients of the same step are compared bit for bit and to
    1e-5 of scale."""
    mp.spawn(_find_phase_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), 'r%d.pt' % r)) for r in (0, 1))
    assert r0['res'] == 'measured' and r1['res'] == 'from the database'
    assert r0['seen'] == r1['seen'] == 2
    assert r0['agree_same'] == r1['agree_same'] == (True, True)
    assert r0['agree_diff'] == r1['agree_diff'] == (False, False)
    assert r0['agree_near'] == r1['agree_near'] == (False, True)          # not the same bits, the same to rounding
    assert r0['agree_shift'] == r1['agree_shift'] == (False, True)
    assert r0['detail_shift']['worst_rel_1d'] > 1e-4 and r0['detail_shift']['worst_rel_weights'] == 0.0
    assert r1['seconds'] >= 0.4                                           # rank 1 waited for rank 0


def test_find_phase_helpers_without_a_process_group():
    from deepipr_amd import distributed as D
    res, _s = D.rank0_first(lambda: 7)
    assert res == 7 and D.ranks_seen(torch.device('cpu')) == 1
    assert D.gradients_agree([torch.ones(2)], torch.device('cpu')) == (None, None)


def _runner_fallback_worker(rank, world, port, out_dir):
    """StepRunner's watch over the first staged replays: ONE rank sees an exchange time-out, BOTH must fall back."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from deepipr_amd import distributed as D
    from deepipr_amd import passport_ops
    from deepipr_amd.experiments.trainer import StepRunner
    D.init_from_env('gloo')

    class FakeKernels:
        def __init__(self):
            self.calls = []
        def sync_timeouts(self):
            return 1 if rank == 1 else 0
        def set_user_sync(self, on):
            self.calls.append(('set_user_sync', on))
        def reset_sync_words(self):
            self.calls.append(('reset_sync_words',))

    class FakeStep:
        built = 0
        def __init__(self):
            FakeStep.built += 1
            self.closed = False
        def __call__(self, data, target):
            with torch.no_grad():
                model.weight.add_(1.0)
                opt.flat_buf.add_(1.0)
            return {'loss': 0.0}
        def describe(self):
            return ''
        def close(self):
            self.closed = True

    class FakeOpt:
        flat_buf = torch.zeros(4)

    fake = FakeKernels()
    real = passport_ops.kernels
    passport_ops.kernels = fake
    try:
        model, opt = torch.nn.Linear(2, 2), FakeOpt()
        w0 = model.weight.detach().clone()
        run = StepRunner(lambda *a: None, model, opt, graph=True)
        run._build = lambda d, t: FakeStep()
        first = run._graphed = run._build(None, None)
        run._probe = [1, {k: v.clone() for k, v in model.state_dict().items()}, opt.flat_buf.clone()]     # as __call__ arms it
        x = torch.zeros(1)
        run._watched(run._graphed(x, x), x, x)
        report = {'calls': fake.calls, 'built': FakeStep.built, 'first_closed': first.closed, 'probe': run._probe,
                  'weight_steps': float((model.weight.detach() - w0).mean()), 'buf': float(opt.flat_buf.mean())}
        with open(os.path.join(out_dir, f'runner{rank}.json'), 'w') as f:
            json.dump(report, f)
    finally:
        passport_ops.kernels = real
        torch.distributed.destroy_process_group()


def test_step_runner_falls_back_on_every_rank_when_one_rank_saw_an_exchange_timeout(tmp_path):
    """the trainer path (not only bench.py) watches the staged step's FIRST replay and, by a MIN all-reduce
    (on host tensors for gloo), takes every rank to the three-launch form together, restoring the state that replay started
    from and running the same batch through the rebuilt step: no batch dropped, no poisoned output handed to the caller."""
    mp.spawn(_runner_fallback_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        r = json.load(open(tmp_path / f'runner{rank}.json'))
        assert r['calls'] == [['set_user_sync', False], ['reset_sync_words']], r
        assert r['built'] == 2 and r['first_closed'] and r['probe'] is None
        # the watched replay undone, one replay of the rebuilt step
        assert r['weight_steps'] == pytest.approx(1.0) and r['buf'] == pytest.approx(1.0)
