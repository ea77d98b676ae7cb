"""Experiment driver behind train_v1.py / train_v23.py.

The reference's harness (experiments/base.py, classification*.py: dataset download, log-dir bookkeeping,
transfer learning) is out of scope; this driver keeps the command-line surface
and does what the hot path needs: build the net from the same flags and JSON configs, install keys
(random / image / shuffle), SGD(momentum .9, wd 1e-4) + MultiStepLR from --lr-config, one process per GPU,
epoch loop through Trainer / TrainerPrivate, history.csv and last.pth.  Data are synthetic tensors of the
dataset's shape (no network / torchvision on the target boxes); a real loader can be passed to run().
"""
import csv
import json
import os
import time

import numpy as np
import torch

from deepipr_amd import distributed as D
from deepipr_amd import passport_generator
from deepipr_amd.experiments.trainer import Trainer
from deepipr_amd.experiments.trainer_private import DualBranch, TrainerPrivate
from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
from deepipr_amd.models.alexnet_normal import AlexNetNormal
from deepipr_amd.models.alexnet_passport import AlexNetPassport
from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
from deepipr_amd.models.resnet_normal import ResNet9, ResNet18
from deepipr_amd.models.resnet_passport import ResNet9Passport, ResNet18Passport
from deepipr_amd.models.resnet_passport_private import ResNet18Private

NUM_CLASSES = {'cifar10': 10, 'cifar100': 100, 'caltech-101': 101, 'caltech-256': 256, 'imagenet1000': 1000}
IMAGE_SIZE = {'cifar10': 32, 'cifar100': 32, 'caltech-101': 32, 'caltech-256': 32, 'imagenet1000': 224}


class SyntheticLoader:
    """Batches of N(0,1) images and uniform labels with the dataset's shape, resident on `device`.
    Has the two things the reference's code needs from a DataLoader: iteration with len(), `.dataset`."""

    class _Dataset:
        def __init__(self, x, y):
            self.x, self.y = x, y

        def __len__(self):
            return self.x.size(0)

        def __getitem__(self, i):
            return self.x[i], self.y[i]

    def __init__(self, samples, batch_size, size, num_classes, device, seed):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(samples, 3, size, size, generator=g).to(device)
        y = torch.randint(0, num_classes, (samples,), generator=g).to(device)
        self.dataset = self._Dataset(x, y)
        self.batch_size = batch_size

    def __len__(self):
        return max(1, len(self.dataset) // self.batch_size)

    def __iter__(self):
        for i in range(len(self)):
            lo = i * self.batch_size
            yield self.dataset.x[lo:lo + self.batch_size], self.dataset.y[lo:lo + self.batch_size]


def scheme_of(args, private):
    """1 = passport (V1), 2 = private passport, 3 = private + backdoor (experiments/base.py:48-55)."""
    if not private:
        return 1 if args['train_passport'] else 0
    return 3 if args['train_backdoor'] else 2


def next_experiment_id(root):
    """Smallest positive integer not yet used as a sub-directory name of `root` (experiments/base.py:76-83)."""
    used = set()
    if os.path.isdir(root):
        used = {int(d) for d in os.listdir(root) if d.isdigit() and os.path.isdir(os.path.join(root, d))}
    i = 1
    while i in used:
        i += 1
    return i


def build_model(args, private, num_classes, device):
    cfg = json.load(open(args['passport_config']))
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': args['norm_type'],
                                              'key_type': args['key_type'], 'sl_ratio': args['sign_loss']})
    arch = args['arch']
    if private:
        model = AlexNetPassportPrivate(3, num_classes, kw) if arch == 'alexnet' else \
            ResNet18Private(num_classes=num_classes, passport_kwargs=kw)
    elif args['train_passport']:
        if arch == 'alexnet':
            model = AlexNetPassport(3, num_classes, kw)
        else:
            model = (ResNet18Passport if arch == 'resnet' else ResNet9Passport)(num_classes=num_classes,
                                                                              passport_kwargs=kw)
    else:
        if arch == 'alexnet':
            model = AlexNetNormal(3, num_classes, args['norm_type'])
        else:
            model = (ResNet18 if arch == 'resnet' else ResNet9)(num_classes=num_classes, norm_type=args['norm_type'])
        if args.get('pretrained_path'):
            model.load_state_dict(torch.load(args['pretrained_path'], map_location='cpu'))
    return model.to(device)


def install_keys(args, model, passport_loader, num_classes, device, example):
    """key_type 'random': lazily drawn on the first forward (done here so that replicas can be synchronised);
    'image' / 'shuffle': 1 / 20 passport images pushed through the plain key-propagation net
    (experiments/classification.py:69-92,130-140)."""
    if args['key_type'] == 'random':
        model.train()
        with torch.no_grad():
            model(example)
        return
    arch = args['arch']
    if arch == 'alexnet':
        plain = AlexNetNormal(3, num_classes, norm_type=args['norm_type'])
    else:
        plain = (ResNet18 if arch == 'resnet' else ResNet9)(num_classes=num_classes, norm_type=args['norm_type'])
    if args.get('pretrained_path'):
        plain.load_state_dict(torch.load(args['pretrained_path'], map_location='cpu'))
    plain = plain.to(device)
    n = 1 if args['key_type'] == 'image' else 20
    key_x, _ = passport_generator.get_key(passport_loader, n)
    key_y, _ = passport_generator.get_key(passport_loader, n)
    passport_generator.set_key(plain, model, key_x.to(device), key_y.to(device))


def run(args, private, train_loader=None, valid_loader=None, wm_loader=None):
    rank, local_rank, world = D.init_from_env(args.get('backend'))
    dev_name = args.get('device') or ('cuda' if torch.cuda.is_available() else None)
    if dev_name is None:
        raise RuntimeError('no GPU: the passport layers run on MI355X only')
    device = torch.device(dev_name, local_rank) if dev_name == 'cuda' else torch.device(dev_name)
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    if args.get('reproducible'):
        from deepipr_amd.reproducible import pin
        pin()                                       # immediate mode, atomic backward-data solver off
    else:
        torch.backends.cudnn.benchmark = True      # MIOpen find mode, as train_v1.py:8
    dataset = args['dataset']
    ncls, size = NUM_CLASSES[dataset], IMAGE_SIZE[dataset]
    lr_config = json.load(open(args['lr_config']))
    epochs = args['epochs'] if args['epochs'] is not None else lr_config['epochs']
    per_gpu = max(1, args['batch_size'] // world)
    samples = args.get('synthetic_samples') or 50 * args['batch_size']
    if train_loader is None:
        train_loader = SyntheticLoader(samples // world, per_gpu, size, ncls, device, seed=1234 + rank)
    if valid_loader is None:
        valid_loader = SyntheticLoader(max(32, per_gpu, samples // (5 * world)), per_gpu, size, ncls, device, seed=4321)
    if wm_loader is None and args['train_backdoor']:
        wm_loader = SyntheticLoader(100, 2, size, ncls, device, seed=99)       # trigger batches of 2 (dataset.py:188-191)

    torch.manual_seed(0)
    np.random.seed(0)
    model = build_model(args, private, ncls, device)
    passport = private or args['train_passport']
    if passport:
        # the passport images come from the validation set, or -- --use-trigger-as-passport -- from the trigger set
        # (experiments/classification.py:37-40: passport_data = prepare_wm(...) instead of valid_data)
        passport_loader = valid_loader
        if args.get('use_trigger_as_passport'):
            passport_loader = wm_loader if wm_loader is not None else \
                SyntheticLoader(100, 2, size, ncls, device, seed=99)      # the trigger set's synthetic stand-in
        install_keys(args, model, passport_loader, ncls, device, next(iter(train_loader))[0])
    # SGD(momentum .9, weight decay 1e-4) as experiments/classification.py:47-50.  Default: FlatSGD (flat buffers,
    # bucketed RCCL all-reduce overlapped with backward, one fused HIP kernel); --ddp: DistributedDataParallel +
    # torch's SGD.
    steps = lr_config[lr_config['type']]
    if args.get('ddp'):
        opt = torch.optim.SGD(model.parameters(), lr=args['lr'], momentum=0.9, weight_decay=0.0001,
                              fused=(device.type == 'cuda'))
        wrap = lambda m: D.replicate(m, device)
    else:
        from deepipr_amd.flat_sgd import FlatSGD
        if passport:
            D.check_keys_materialised(model)
        D.broadcast_state(model, 0)
        opt = FlatSGD(model.parameters(), lr=args['lr'], momentum=0.9, weight_decay=0.0001)
        wrap = lambda m: m
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, steps, lr_config['gamma']) if len(steps) else None
    # hipGraph replay is the default on the GPU (--eager switches it off): at the reference's batch sizes the eager step is
    # host-bound (~250-500 dispatches through Python autograd: config R +7 %, the 32-image shard of config P +46 %,
    # DESIGN.md 5); a replayed one is not.  One GPU: the whole step is one graph; several: the staged step
    # (experiments/staged.py).  Ragged last batches run the eager step.  DDP's own hooks cannot be captured.
    graph = device.type == 'cuda' and not args.get('eager') and not args.get('ddp')
    if private:
        net = wrap(DualBranch(model))
        trainer = TrainerPrivate(net, opt, sched, device, graph=graph)
    else:
        net = wrap(model)
        trainer = Trainer(net, opt, sched, device, graph=graph)

    scheme = scheme_of(args, private)
    # logs/<arch>_<dataset>_v<scheme>[_tag]/<id>/{config.json, history.csv, models/{best,last,epoch-N}.pth}
    # (experiments/base.py:57-74,110-150): training takes the smallest unused id, --eval reads --exp-id's best.pth
    root = os.path.join(args.get('logdir') or 'logs', '%s_%s_v%d%s' % (
        args['arch'], dataset, scheme, '_' + args['tag'] if args.get('tag') else ''))
    if args.get('eval'):
        logdir = os.path.join(root, str(args['exp_id']))
        path = os.path.join(logdir, 'models', 'best.pth')
        if os.path.exists(path):
            model.load_state_dict(torch.load(path, map_location='cpu'))
        else:
            print('Warning: No such Experiment -> %s' % path)
        return trainer.test(valid_loader)
    exp_id = [next_experiment_id(root) if rank == 0 else 0]
    if world > 1:
        torch.distributed.broadcast_object_list(exp_id, src=0)
    logdir = os.path.join(root, str(exp_id[0]))
    history = []
    best_acc = float('-inf')
    if rank == 0:
        os.makedirs(os.path.join(logdir, 'models'), exist_ok=True)
        json.dump({k: v for k, v in args.items()}, open(os.path.join(logdir, 'config.json'), 'w'), indent=4)
    for ep in range(1, epochs + 1):
        t0 = time.time()
        row = {'epoch': ep}
        row.update({'train_' + k: v for k, v in trainer.train(ep, train_loader, wm_loader).items()})
        if rank == 0:
            row.update({'valid_' + k: v for k, v in trainer.test(valid_loader).items()})
            if wm_loader is not None:
                row.update({'wm_' + k: v for k, v in trainer.test(wm_loader, 'WM Result').items()})
            n_img = len(train_loader) * per_gpu * world
            row['train_img_per_s'] = n_img / max(1e-9, row['train_time'])
            history.append(row)
            cols = sorted({c for r in history for c in r})
            with open(os.path.join(logdir, 'history.csv'), 'w', newline='') as f:
                w = csv.DictWriter(f, cols)
                w.writeheader()
                w.writerows(history)
            sd = {k: v.cpu() for k, v in model.state_dict().items()}
            # best.pth follows valid_acc (classification.py:298-301) / valid_total_acc (classification_private.py:150-153)
            score = row.get('valid_total_acc' if private else 'valid_acc', float('-inf'))
            if score > best_acc:
                best_acc = score
                torch.save(sd, os.path.join(logdir, 'models', 'best.pth'))
            torch.save(sd, os.path.join(logdir, 'models', 'last.pth'))
            if args['save_interval'] and ep % args['save_interval'] == 0:
                torch.save(sd, os.path.join(logdir, 'models', 'epoch-%d.pth' % ep))
            print('epoch %d done in %.2fs: %s' % (ep, time.time() - t0, {k: round(v, 4) if isinstance(v, float)
                                                                         else v for k, v in row.items()}))
        D.barrier()
    return {'logdir': logdir, 'history': history, 'model': model}
