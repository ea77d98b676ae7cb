#!/usr/bin/env python
"""Scheme V1 entry point -- same flags as the reference's train_v1.py:11-87.

    python train_v1.py --arch resnet --train-passport --passport-config passport_configs/resnet18_passport.json
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_v1.py ...   (one process / GPU)
"""
import sys
from pprint import pprint

from deepipr_amd.experiments.cli import make_parser
from deepipr_amd.experiments.runner import run


def main(argv=None):
    args = vars(make_parser(private=False).parse_args(argv))
    pprint(args)
    if args['transfer_learning']:
        raise SystemExit('transfer learning is outside this build\'s scope (SURVEY.md 2, row 12)')
    out = run(args, private=False)
    print('Training done at', out.get('logdir') if isinstance(out, dict) else out)
    return out


if __name__ == '__main__':
    main(sys.argv[1:])
