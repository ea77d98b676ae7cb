"""Scheme-V1 train / test loops -- drop-in for the reference's experiments/trainer.py:46-214.

Same constructor and return dictionaries.  What differs is how the step is driven on an MI355X:
  * one process per GPU; a DistributedDataParallel-wrapped model (RCCL all-reduce over xGMI) is used
    as is -- the reference's nn.DataParallel (trainer.py:92-93) is never applied;
  * the three per-step .item() host syncs (trainer.py:147-149) are replaced by on-device meters that
    are read once per epoch (or every `log_interval` batches when progress printing is wanted).
"""
import time

import torch
import torch.nn.functional as F

from deepipr_amd.models.losses.sign_loss import SignLoss
from deepipr_amd.passport_ops import cross_entropy_top1, scalar_sums


def accuracy(output, target, topk=(1,)):
    """precision@k in percent (trainer.py:28-43)."""
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.view(1, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def sign_loss_modules(model):
    return [m for m in model.modules() if isinstance(m, SignLoss)]


def reset_sign_losses(model):
    for m in sign_loss_modules(model):
        m.reset()


def total_sign_loss(model, device):
    """Sum of the SignLoss modules' losses (trainer.py:140-142) -- without a zero-filled start value and its extra add."""
    total = None
    for m in sign_loss_modules(model):
        if isinstance(m.loss, torch.Tensor):
            total = m.loss if total is None else total + m.loss
    return total if total is not None else torch.zeros((), device=device)


def sign_loss_terms(model):
    """The SignLoss modules' losses of the last forward pass, in module order (the order total_sign_loss adds them in)."""
    return [m.loss for m in sign_loss_modules(model) if isinstance(m.loss, torch.Tensor)]


def mean_sign_acc(model, device):
    """Mean over SignLoss modules of the accuracy of the LAST forward (the meters are reset every step,
    trainer.py:131-133,162-171)."""
    mods = sign_loss_modules(model)
    acc = torch.zeros((), device=device)
    for m in mods:
        acc = acc + m.acc
    return acc / len(mods) if mods else acc


def next_trigger_batch(state, wm_dataloader):
    """Cycle through the trigger-set loader (trainer.py:115-120)."""
    try:
        return next(state['it'])
    except StopIteration:
        state['it'] = iter(wm_dataloader)
        return next(state['it'])


def _check_exchange(device):
    """Once per epoch: an in-launch partial-sum exchange of the single-pass norm kernels that ever timed out (its
    outputs were poisoned with NaN) must stop the run, not train on."""
    if torch.device(device).type == 'cuda':
        from deepipr_amd import passport_ops
        passport_ops.kernels.check_exchange()


def forward_loss_v1(model, data, target):
    """Forward half of the V1 step (trainer.py:131-142): -> (objective = CE + sum of sign losses, the step's device
    scalars (loss, sign_loss, top-1 %)).  Separate from train_step_v1 so that a driver can run the backward pass
    itself -- experiments/staged.py back-propagates stage by stage to overlap the gradient exchange."""
    reset_sign_losses(model)
    pred = model(data)
    loss, top1 = cross_entropy_top1(pred, target)          # F.cross_entropy + accuracy()[0], one fused launch on the GPU
    _ce, sign_loss, objective = scalar_sums([loss], sign_loss_terms(model))      # sign_loss += m.loss ...; loss + sign_loss: ONE launch
    if sign_loss is None:
        sign_loss = torch.zeros((), device=data.device)
    return objective, (loss.detach(), sign_loss.detach(), top1)


def train_step_v1(model, optimizer, data, target):
    """One batch: zero_grad, reset sign losses, forward, CE + sum of sign losses, backward, SGD step
    (trainer.py:128-145).  Returns device scalars (loss, sign_loss, top-1 %), no host sync."""
    optimizer.zero_grad(set_to_none=True)
    objective, out = forward_loss_v1(model, data, target)
    objective.backward()
    optimizer.step()
    return out


train_step_v1.forward_loss = forward_loss_v1


class Tester(object):
    """graph=True: every batch is one hipGraph replay (experiments/graph_step.GraphedEval, one graph per batch shape) instead
    of an eager, host-bound forward; same numbers (the same kernels run)."""

    def __init__(self, model, device, verbose=True, graph=False):
        self.model = model
        self.device = device
        self.verbose = verbose
        self.graph = graph and torch.device(device).type == 'cuda'
        self._batch = None

    def _run(self, data, target):
        from deepipr_amd.experiments.graph_step import GraphedEval
        if not self.graph:
            return GraphedEval.batch(self.model, data, target)
        if self._batch is None:
            self._batch = GraphedEval(self.model)
        return self._batch(data, target)

    def test(self, dataloader, msg='Testing Result', compare=[]):
        self.model.eval()
        start = time.time()
        loss_sum = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device)
        count = 0
        with torch.no_grad():
            for load in dataloader:
                data = load[0].to(self.device, non_blocking=True)
                target = load[1].to(self.device, non_blocking=True)
                top, loss, hits = self._run(data, target)
                loss_sum += loss
                compare.append((top.clone() if self.graph else top, target))
                correct += hits
                count += data.size(0)
        loss = loss_sum.item() / count
        acc = 100 * correct.item() / count
        if self.verbose:
            print(f'{msg}: Loss: {loss:6.4f} Acc: {acc:6.2f} ({time.time() - start:.2f}s)')
            print()
        return {'loss': loss, 'acc': acc, 'time': time.time() - start}


class StepRunner:
    """Runs step_fn eagerly, or -- graph=True -- from hipGraphs captured on the first batch of each shape (one GPU: up to
    MAX_SHAPES further shapes, i.e. the ragged last batch too; data parallel: the first shape, others take the eager step).
      one GPU      : the whole step, optimiser included, is one graph (experiments/graph_step.py);
      data parallel: the staged form (experiments/staged.py) -- forward + backward as a few graphs cut at the gradient
                     buckets' boundaries, each bucket's RCCL all-reduce launched between the replays, one fused SGD."""

    MAX_SHAPES = 3

    def __init__(self, step_fn, model, optimizer, graph=False):
        self.step_fn, self.model, self.optimizer = step_fn, model, optimizer
        self.graph = graph
        self.distributed = torch.distributed.is_available() and torch.distributed.is_initialized()
        if self.distributed and torch.distributed.get_world_size() > 1:
            from deepipr_amd import passport_ops
            passport_ops.prefer_own_kernels()                  # every rank on the same, bit-reproducible kernels
        self._graphed = None
        self._shape = None
        self._others = {}                  # one GPU: (data shape, target shape) -> graph of a further batch shape
        self._probe = None                 # staged data-parallel step: (replays left to watch, state to fall back to)

    def _build(self, data, target):
        if self.distributed and hasattr(self.optimizer, 'configure_stages') and hasattr(self.step_fn, 'forward_loss'):
            from deepipr_amd.experiments.staged import StagedStep
            return StagedStep(self.step_fn, self.model, self.optimizer, data, target, graph=True, warmup=0)
        from deepipr_amd.experiments.graph_step import GraphedTrainStep
        return GraphedTrainStep(self.step_fn, self.model, self.optimizer, data, target, warmup=0,
                                optimizer_in_graph=not self.distributed)

    def __call__(self, data, target):
        if not self.graph or not data.is_cuda:
            return self.step_fn(self.model, self.optimizer, data, target)
        if self._graphed is None:
            self._graphed = self._build(data, target)
            self._shape = (tuple(data.shape), tuple(target.shape))
            if self.distributed and hasattr(self._graphed, 'describe') and hasattr(self.optimizer, 'flat_buf'):
                # The staged step lets collectives overlap the split-channel single-pass kernels (policy "shared",
                # staged.py); should their in-launch exchange ever time out next to one, the first replays show it.
                # Watch the FIRST one, all ranks together, instead of finding NaN statistics at the end of the epoch: checked
                # right after it, a time-out costs nothing but this one batch run twice -- no batch is dropped and no
                # poisoned output reaches the caller's meters (three watched replays lost two batches).
                self._probe = [1, {k: v.clone() for k, v in self.model.state_dict().items()}, self.optimizer.flat_buf.clone()]
            return self._watched(self._graphed(data, target), data, target)      # capture does not execute: replay the first batch
        key = (tuple(data.shape), tuple(target.shape))
        if key != self._shape:
            # another batch shape (the ragged last batch of an epoch): on one GPU it gets a graph of its own -- FlatSGD keeps
            # a chunk table per capture, the norm layers' statistics and the parameters are the same tensors -- instead of the
            # host-bound eager step (2-3x the replayed step's time); a data-parallel run keeps the eager step for it (the staged
            # form's buckets and events belong to one shape)
            if self.distributed or len(self._others) >= self.MAX_SHAPES and key not in self._others:
                return self.step_fn(self.model, self.optimizer, data, target)
            g = self._others.get(key)
            if g is None:
                g = self._others[key] = self._build(data, target)
            return g(data, target)
        return self._watched(self._graphed(data, target), data, target)

    def _watched(self, out, data, target):
        if self._probe is None:
            return out
        self._probe[0] -= 1
        if self._probe[0] > 0:
            return out
        from deepipr_amd import passport_ops
        kernels = passport_ops.kernels
        if data.is_cuda:
            torch.cuda.synchronize()
        # every rank takes the same form; a gloo group reduces host tensors (as distributed.ranks_seen does)
        on_host = torch.distributed.get_backend() == 'gloo'
        ok = torch.tensor([0.0 if kernels.sync_timeouts() else 1.0], device='cpu' if on_host else data.device)
        torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        _n, state, flat = self._probe
        self._probe = None
        if float(ok.item()) > 0.5:
            return out
        print('deepipr_amd: an in-launch exchange of the single-pass norm kernels timed out next to a collective on some '
              'rank; all ranks fall back to the three-launch form for the split-channel layers, restore the state from '
              'before this step, capture the step again and run this batch through it', flush=True)
        kernels.set_user_sync(False)
        kernels.reset_sync_words()
        if hasattr(self._graphed, 'close'):
            self._graphed.close()
        with torch.no_grad():
            self.model.load_state_dict(state)
            self.optimizer.flat_buf.copy_(flat)
        self._graphed = self._build(data, target)
        return self._graphed(data, target)


class Trainer(object):
    def __init__(self, model, optimizer, scheduler, device, log_interval=0, graph=False):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.device = device
        self.log_interval = log_interval
        self.step = StepRunner(train_step_v1, model, optimizer, graph)
        self.tester = Tester(model, device, verbose=False, graph=graph)

    def train(self, e, dataloader, wm_dataloader=None):
        self.model.train()
        dev = self.device
        meters = torch.zeros(3, device=dev)                     # sign loss, loss, acc
        wm_state = {'it': iter(wm_dataloader)} if wm_dataloader is not None else None
        start = time.time()
        for i, (data, target) in enumerate(dataloader):
            data = data.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)
            if wm_state is not None:                            # V1 + backdoor: append the trigger pair
                wm_data, wm_target = next_trigger_batch(wm_state, wm_dataloader)
                data = torch.cat([data, wm_data.to(dev, non_blocking=True)], dim=0)
                target = torch.cat([target, wm_target.to(dev, non_blocking=True)], dim=0)
            loss, sign_loss, acc = self.step(data, target)
            meters += torch.stack([sign_loss, loss, acc])
            if self.log_interval and (i + 1) % self.log_interval == 0:
                s, l, a = (meters / (i + 1)).tolist()
                print(f'Epoch {e:3d} [{i:4d}/{len(dataloader):4d}] Sign Loss: {s:6.4f} Loss: {l:6.4f} '
                      f'Acc: {a:.4f} ({time.time() - start:.2f}s)', end='\r')
        if self.log_interval:
            print()
        n = max(1, len(dataloader))
        sign_acc = mean_sign_acc(self.model, dev)
        s, l, a, sa = torch.cat([meters / n, sign_acc.reshape(1)]).tolist()     # the epoch's only host sync
        _check_exchange(dev)
        if self.scheduler is not None:
            self.scheduler.step()
        return {'loss': l, 'sign_loss': s, 'sign_acc': sa, 'acc': a, 'time': time.time() - start}

    def test(self, dataloader, msg='Testing Result'):
        out = self.tester.test(dataloader, msg, compare=[])
        print(f'{msg}: Loss: {out["loss"]:6.4f} Acc: {out["acc"]:6.2f} ({out["time"]:.2f}s)')
        print()
        return out
