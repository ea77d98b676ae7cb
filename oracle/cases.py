"""Shared definitions of the golden-fixture cases (TEST INFRASTRUCTURE, see oracle/__init__.py).

tools/gen_golden.py runs each case on the real reference and stores its outputs in
tests/golden/<case>.npz; the tests re-run the same case on the oracle (CPU) and on the HIP build
(GPU) and compare.  Passport configs below are the trees of the reference's
passport_configs/alexnet_passport.json:1-7 and passport_configs/resnet18_passport.json:1-46.
"""
import copy


def alexnet_config(l4=True, l5=True, l6=True):
    return {'0': False, '2': False, '4': l4, '5': l5, '6': l6}


def resnet18_config(layer4=True):
    def blk(flag, shortcut):
        d = {'convbnrelu_1': flag, 'convbn_2': flag}
        if shortcut:
            d['shortcut'] = flag
        return d
    cfg = {'convbnrelu_1': False}
    for li in (1, 2, 3, 4):
        flag = layer4 if li == 4 else False
        cfg['layer%d' % li] = {'0': blk(flag, li != 1), '1': blk(flag, False)}
    return cfg


SGD = dict(lr=0.01, momentum=0.9, weight_decay=1e-4)   # experiments/classification.py:47-50

_SIG = copy.deepcopy(alexnet_config())
_SIG['4'] = 'this is my signature'                      # 20 chars = 160 bits <= 384 channels
_SIG['6'] = 'DeepIPR'

CASES = {
    # name            arch       scheme norm   ncls  n  config
    'alexnet_v1':    dict(arch='alexnet', scheme=1, norm='bn', ncls=10, n=4, config=alexnet_config()),
    'alexnet_v1_gn': dict(arch='alexnet', scheme=1, norm='gn', ncls=10, n=4, config=alexnet_config()),
    'alexnet_v1_in': dict(arch='alexnet', scheme=1, norm='in', ncls=10, n=4, config=alexnet_config()),
    'alexnet_v1_none': dict(arch='alexnet', scheme=1, norm='none', ncls=10, n=4, config=alexnet_config()),
    'alexnet_v1_sig': dict(arch='alexnet', scheme=1, norm='bn', ncls=10, n=4, config=_SIG),
    'alexnet_v2':    dict(arch='alexnet', scheme=2, norm='bn', ncls=100, n=4, config=alexnet_config()),
    'resnet18_v1':   dict(arch='resnet18', scheme=1, norm='bn', ncls=10, n=4, config=resnet18_config()),
    'resnet18_v1_wm': dict(arch='resnet18', scheme=1, norm='bn', ncls=10, n=4, config=resnet18_config(), wm=True),
    'resnet18_v2':   dict(arch='resnet18', scheme=2, norm='bn', ncls=100, n=4, config=resnet18_config()),
    'resnet18_v3':   dict(arch='resnet18', scheme=3, norm='bn', ncls=100, n=6, config=resnet18_config(), wm=True),
}

ALPHA = 0.1                                             # train_v1.py:33 --sign-loss default
