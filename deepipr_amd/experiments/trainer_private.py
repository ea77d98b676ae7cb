"""Scheme V2 / V3 train / test loops and signature detection -- drop-in for the reference's
experiments/trainer_private.py:29-257.

A step is two forwards (ind=0 public, ind=1 private) whose cross-entropies are summed, plus the private
sign losses, and ONE backward (trainer_private.py:159-173).  Under DistributedDataParallel the two
forwards are issued through `DualBranch`, so the wrapper sees a single forward per backward.
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from deepipr_amd.experiments.trainer import (StepRunner, _check_exchange, accuracy, cross_entropy_top1, mean_sign_acc,
                                             next_trigger_batch,
                                             reset_sign_losses, scalar_sums, sign_loss_terms, total_sign_loss)
from deepipr_amd.models.layers.passportconv2d import PassportBlock
from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock


class DualBranch(nn.Module):
    """model(x, ind=0) and model(x, ind=1) in one forward call, public branch first so that the batch-norm
    running statistics are updated in the reference's order.  Nets that offer forward_dual run the layers in front of
    their first private passport layer once for both branches (models/_builders.shared_trunk: same outputs, same
    gradients, same running statistics; DEEPIPR_NO_SHARED_TRUNK=1 restores the two full passes).  A net that carries
    hooks the one-pass form would not fire as two calls do -- forward pre-hooks, backward (pre-)hooks, global hooks, or
    any hook on a submodule (feature collectors of the attack / fine-tune scripts) -- takes the two full passes."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, data):
        m = self.model
        from torch.nn.modules import module as _mod
        if (not hasattr(m, 'forward_dual') or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks
                or _mod._global_forward_hooks or _mod._global_forward_pre_hooks or _mod._global_backward_hooks
                or _mod._global_backward_pre_hooks or _submodule_hooks(m)):
            return m(data, ind=0), m(data, ind=1)
        outs = list(m.forward_dual(data))
        # the net's own forward hooks see two calls, as with model(data, ind=0); model(data, ind=1)
        for ind, out in enumerate(outs):
            for hid, hook in list(m._forward_hooks.items()):
                if hid in m._forward_hooks_with_kwargs:
                    res = hook(m, (data,), {'ind': ind}, out)
                else:
                    res = hook(m, (data,), out)
                if res is not None:
                    out = res
            outs[ind] = out
        return outs[0], outs[1]


def _submodule_hooks(model):
    """Any hook on a module below the net itself (the net's own forward hooks are re-fired by DualBranch)."""
    for sub in model.modules():
        if sub is not model and (sub._forward_hooks or sub._forward_pre_hooks or sub._backward_hooks
                                 or sub._backward_pre_hooks):
            return True
    return False


def _unwrap(model):
    while hasattr(model, 'module'):
        model = model.module
    return model.model if isinstance(model, DualBranch) else model


def forward_loss_v23(dual, data, target):
    """Forward half of the V2 / V3 step (trainer_private.py:159-171): both branches, -> (objective, device scalars
    (loss, sign_loss, public top-1 %, private top-1 %)); see trainer.forward_loss_v1."""
    reset_sign_losses(dual)
    pred_public, pred_private = dual(data)
    loss_public, top1_public = cross_entropy_top1(pred_public, target)
    loss_private, top1_private = cross_entropy_top1(pred_private, target)
    loss, sign_loss, objective = scalar_sums([loss_public, loss_private], sign_loss_terms(dual))     # the three sums: ONE launch
    if sign_loss is None:
        sign_loss = torch.zeros((), device=data.device)
    return objective, (loss.detach(), sign_loss.detach(), top1_public, top1_private)


def train_step_v23(dual, optimizer, data, target):
    """-> device scalars (loss, sign_loss, public top-1 %, private top-1 %)."""
    optimizer.zero_grad(set_to_none=True)
    objective, out = forward_loss_v23(dual, data, target)
    objective.backward()
    optimizer.step()
    return out


train_step_v23.forward_loss = forward_loss_v23


class TesterPrivate(object):
    """graph=True: every test batch is one hipGraph replay per (branch, batch shape) -- experiments/graph_step.GraphedEval."""

    def __init__(self, model, device, verbose=True, graph=False):
        self.model = model
        self.device = device
        self.verbose = verbose
        self.graph = graph and torch.device(device).type == 'cuda'
        self._batch = {}

    def _run(self, model, ind, data, target):
        from deepipr_amd.experiments.graph_step import GraphedEval
        if not self.graph:
            return GraphedEval.batch(lambda d: model(d, ind=ind), data, target)
        if ind not in self._batch:
            self._batch[ind] = GraphedEval(model, forward=lambda d: model(d, ind=ind))
        return self._batch[ind](data, target)

    def test_signature(self):
        """sign(gamma) == b per passport layer (trainer_private.py:37-71): 'private_<name>' for
        PassportPrivateBlock (gamma of the private branch), 'public_<name>' for PassportBlock."""
        model = _unwrap(self.model)
        model.eval()
        names, rates = [], []
        with torch.no_grad():
            for name, m in model.named_modules():
                if isinstance(m, PassportPrivateBlock):
                    bits = m.get_scale(ind=1).view(-1).sign()
                    names.append('private_' + name)
                elif isinstance(m, PassportBlock):
                    bits = m.get_scale().view(-1).sign()
                    names.append('public_' + name)
                else:
                    continue
                rates.append((bits == m.b).float().mean())
        values = torch.stack(rates).tolist() if rates else []
        res = dict(zip(names, values))
        for kind in ('private', 'public'):
            sel = [v for k, v in res.items() if k.startswith(kind + '_')]
            if sel and self.verbose:
                print(f'{kind.capitalize()} Sign Detection Accuracy: {sum(sel) / len(sel) * 100:6.4f}')
        return res

    def test(self, dataloader, msg='Testing Result', ind=0):
        model = _unwrap(self.model)
        model.eval()
        start = time.time()
        loss_sum = torch.zeros((), device=self.device)
        correct = torch.zeros((), device=self.device)
        count = 0
        with torch.no_grad():
            for load in dataloader:
                data = load[0].to(self.device, non_blocking=True)
                target = load[1].to(self.device, non_blocking=True)
                _top, loss, hits = self._run(model, ind, data, target)
                loss_sum += loss
                correct += hits
                count += data.size(0)
        loss = loss_sum.item() / count
        acc = 100 * correct.item() / count
        if self.verbose:
            print(f'{msg}: Loss: {loss:6.4f} Acc: {acc:6.2f} ({time.time() - start:.2f}s)')
            print()
        return {'loss': loss, 'acc': acc, 'time': time.time() - start}


class TrainerPrivate(object):
    def __init__(self, model, optimizer, scheduler, device, log_interval=0, graph=False):
        self.model = model
        self.dual = model if isinstance(_strip_ddp(model), DualBranch) else DualBranch(model)
        self.step = StepRunner(train_step_v23, self.dual, optimizer, graph)
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.device = device
        self.log_interval = log_interval
        self.tester = TesterPrivate(model, device)
        self.quiet_tester = TesterPrivate(model, device, verbose=False, graph=graph)

    def train(self, e, dataloader, wm_dataloader=None):
        self.dual.train()
        dev = self.device
        meters = torch.zeros(4, device=dev)                     # loss, sign loss, public acc, private acc
        wm_state = {'it': iter(wm_dataloader)} if wm_dataloader is not None else None
        start = time.time()
        for i, (data, target) in enumerate(dataloader):
            data = data.to(dev, non_blocking=True)
            target = target.to(dev, non_blocking=True)
            if wm_state is not None:                            # V3: trigger-set pair appended to the batch
                wm_data, wm_target = next_trigger_batch(wm_state, wm_dataloader)
                data = torch.cat([data, wm_data.to(dev, non_blocking=True)], dim=0)
                target = torch.cat([target, wm_target.to(dev, non_blocking=True)], dim=0)
            meters += torch.stack(self.step(data, target))
            if self.log_interval and (i + 1) % self.log_interval == 0:
                l, s, pu, pr = (meters / (i + 1)).tolist()
                print(f'Epoch {e:3d} [{i:4d}/{len(dataloader):4d}] Loss: {l:6.4f} Sign Loss: {s:6.4f} '
                      f'Priv. Acc: {pr:.4f} Publ. Acc: {pu:.4f} ({time.time() - start:.2f}s)', end='\r')
        if self.log_interval:
            print()
        n = max(1, len(dataloader))
        sign_acc = mean_sign_acc(self.dual, dev)
        raw = torch.cat([meters, sign_acc.reshape(1)]).tolist()                  # the epoch's only host sync
        _check_exchange(dev)
        if self.scheduler is not None:
            self.scheduler.step()
        # the reference divides every meter by len(dataloader) except sign_loss (trainer_private.py:189-191)
        return {'loss': raw[0] / n, 'sign_loss': raw[1], 'sign_acc': raw[4],
                'acc_public': raw[2] / n, 'acc_private': raw[3] / n, 'time': time.time() - start}

    def test(self, dataloader, msg='Testing Result'):
        out = {}
        quiet = self.quiet_tester
        for i, key in enumerate(('public', 'private')):
            r = quiet.test(dataloader, msg, ind=i)
            print(f'{msg} {key}: Loss: {r["loss"]:6.4f} Acc: {r["acc"]:6.2f} ({r["time"]:.2f}s)')
            print()
            out.update({'loss_' + key: r['loss'], 'acc_' + key: r['acc'], 'time_' + key: r['time']})
        out['total_acc'] = (out['acc_public'] + out['acc_private']) / 2
        print(f'Total acc: {out["total_acc"]:.2f}')
        print()
        for key, val in self.tester.test_signature().items():
            out['s_' + key] = val
        return out


def _strip_ddp(model):
    while hasattr(model, 'module'):
        model = model.module
    return model
