"""Argument parser shared by train_v1.py / train_v23.py: the reference's flags (train_v1.py:12-74,
train_v23.py:12-74) plus a few for synthetic data and device selection."""
import argparse

DATASETS = ['cifar10', 'cifar100', 'caltech-101', 'caltech-256', 'imagenet1000']


def make_parser(private):
    p = argparse.ArgumentParser()
    p.add_argument('--arch', default='alexnet', choices=['alexnet', 'resnet'] + ([] if private else ['resnet9']),
                   help='architecture (default: alexnet)')
    p.add_argument('--batch-size', type=int, default=64, help='global batch size (default: 64)')
    p.add_argument('--epochs', type=int, default=200, help='training epochs (default: 200)')
    p.add_argument('--lr', type=float, default=0.01, help='learning rate (default: 0.01)')
    p.add_argument('--dataset', default='cifar10', choices=DATASETS, help='dataset whose shape / classes are used')
    p.add_argument('--norm-type', default='bn', choices=['bn', 'gn', 'in', 'none'], help='norm type (default: bn)')
    # passport arguments
    p.add_argument('--key-type', choices=['random', 'image', 'shuffle'], default='shuffle',
                   help='passport key type (default: shuffle)')
    p.add_argument('--sign-loss', type=float, default=0.1, help='sign loss weight alpha (default: 0.1)')
    p.add_argument('--use-trigger-as-passport', action='store_true', default=False)
    p.add_argument('--train-passport', action='store_true', default=False, help='train passport (V1)')
    p.add_argument('--train-backdoor', action='store_true', default=False, help='add trigger-set images (V3)')
    p.add_argument('--train-private', action='store_true', default=private, help='train private (V2/V3)')
    # paths
    p.add_argument('--pretrained-path', help='state_dict of the plain key-propagation / baseline net')
    p.add_argument('--lr-config', default='lr_configs/default.json')
    p.add_argument('--passport-config', default='passport_configs/alexnet_passport.json')
    # misc
    p.add_argument('--save-interval', type=int, default=0)
    p.add_argument('--eval', action='store_true', default=False)
    p.add_argument('--exp-id', type=int, default=1)
    p.add_argument('--tag')
    # transfer learning flags are accepted for command-line compatibility; the TL harness is out of scope
    p.add_argument('--transfer-learning', action='store_true', default=False)
    p.add_argument('--tl-dataset', default='cifar100', choices=DATASETS)
    p.add_argument('--tl-scheme', default='rtal', choices=['rtal', 'ftal'])
    # additions
    p.add_argument('--synthetic-samples', type=int, default=0, help='synthetic training samples per epoch')
    p.add_argument('--device', default=None, help='cuda (default) | cpu (tests only: needs patched kernels)')
    p.add_argument('--backend', default=None, help='torch.distributed backend (default nccl = RCCL)')
    p.add_argument('--logdir', default='logs')
    p.add_argument('--ddp', action='store_true', default=False,
                   help='DistributedDataParallel + torch SGD instead of the default FlatSGD data parallelism')
    p.add_argument('--graph', action='store_true', default=False,
                   help='(the default on the GPU; kept for compatibility) replay the train step from captured hipGraphs: '
                        'one GPU: the whole step; several: forward + backward in stages, the gradient buckets all-reduced '
                        'between the replays, one fused SGD kernel')
    p.add_argument('--reproducible', action='store_true', default=False,
                   help='bit-reproducible steps: MIOpen in immediate mode with its atomic backward-data solver switched '
                        'off (deepipr_amd/reproducible.py) instead of find mode')
    p.add_argument('--eager', action='store_true', default=False,
                   help='eager dispatch instead of hipGraph replay (several GPUs: gradient exchange launched from '
                        'gradient hooks, overlapped with backward)')
    return p
