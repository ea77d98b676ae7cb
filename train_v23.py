#!/usr/bin/env python
"""Scheme V2 / V3 entry point -- same flags as the reference's train_v23.py:11-87 (--train-private is always
on; --train-backdoor selects V3).

    python train_v23.py --arch resnet --passport-config passport_configs/resnet18_passport.json --dataset cifar100
"""
import sys
from pprint import pprint

from deepipr_amd.experiments.cli import make_parser
from deepipr_amd.experiments.runner import run


def main(argv=None):
    args = vars(make_parser(private=True).parse_args(argv))
    pprint(args)
    if args['transfer_learning']:
        raise SystemExit('transfer learning is outside this build\'s scope (SURVEY.md 2, row 12)')
    out = run(args, private=True)
    print('Training done at', out.get('logdir') if isinstance(out, dict) else out)
    return out


if __name__ == '__main__':
    main(sys.argv[1:])
