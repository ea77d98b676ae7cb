"""Shared definitions of the golden-fixture cases (TEST INFRASTRUCTURE, see oracle/__init__.py).

tools/gen_golden.py runs each case on the real reference and stores its outputs in
tests/golden/<case>.npz; the tests re-run the same case on the oracle (CPU) and on the HIP build
(GPU) and compare.  Passport configs below are the trees of the reference's
passport_configs/alexnet_passport.json:1-7 and passport_configs/resnet18_passport.json:1-46.
"""
import copy

import numpy as np


def alexnet_config(l4=True, l5=True, l6=True):
    return {'0': False, '2': False, '4': l4, '5': l5, '6': l6}


def resnet18_config(layer4=True, blocks=2):
    def blk(flag, shortcut):
        d = {'convbnrelu_1': flag, 'convbn_2': flag}
        if shortcut:
            d['shortcut'] = flag
        return d
    cfg = {'convbnrelu_1': False}
    for li in (1, 2, 3, 4):
        flag = layer4 if li == 4 else False
        cfg['layer%d' % li] = {'0': blk(flag, li != 1)}
        if blocks > 1:
            cfg['layer%d' % li]['1'] = blk(flag, False)
    return cfg


def resnet9_config(layer4=True):
    """passport_configs/resnet9_passport.json: one BasicBlock per layer (models/resnet_passport.py:187-188)."""
    return resnet18_config(layer4, blocks=1)


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
    # the other norm types at ResNet shapes (models/layers/passportconv2d.py:59-62: GroupNorm(o // 16, o) /
    # InstanceNorm2d(o), both affine=False, inside the passport layers; affine GroupNorm / InstanceNorm ConvBlocks)
    'resnet18_v1_gn': dict(arch='resnet18', scheme=1, norm='gn', ncls=10, n=4, config=resnet18_config()),
    'resnet18_v2_in': dict(arch='resnet18', scheme=2, norm='in', ncls=100, n=4, config=resnet18_config()),
    # ImageNet geometry: num_classes == 1000 switches the 7x7/2 stem + 3x3/2 max-pool on
    # (models/resnet_passport.py:94-98); 3x224x224 inputs, layer4 passports act on 7x7 maps
    'resnet18_v1_imagenet': dict(arch='resnet18', scheme=1, norm='bn', ncls=1000, n=2, hw=224,
                                 config=resnet18_config()),
    # --key-type shuffle, the default of train_v1.py:30-31: keys are activations of a plain net
    # (passport_generator.py:30-43 -> set_intermediate_keys, models/resnet_passport.py:32-65,145-161;
    # models/alexnet_passport.py:104-112), reduced from `nkeys` candidates per layer by passport_selection
    'resnet18_v1_shuffle': dict(arch='resnet18', scheme=1, norm='bn', ncls=10, n=4, config=resnet18_config(),
                                key_type='shuffle', nkeys=4),
    'alexnet_v1_shuffle': dict(arch='alexnet', scheme=1, norm='bn', ncls=10, n=4, config=alexnet_config(),
                               key_type='shuffle', nkeys=5),
    # --key-type image (train_v1.py:30): ONE passport image per key, pushed through the plain net; set_key keeps a
    # batch-1 key as it is (passportconv2d.py:125-137: no passport_selection), passport_generator.py:30-43
    'resnet18_v1_image': dict(arch='resnet18', scheme=1, norm='bn', ncls=10, n=4, config=resnet18_config(),
                              key_type='image', nkeys=1),
    # ResNet9 (--arch resnet9, models/resnet_passport.py:187-188: BasicPassportBlock x [1, 1, 1, 1])
    'resnet9_v1':    dict(arch='resnet9', scheme=1, norm='bn', ncls=10, n=4, config=resnet9_config()),
}

ALPHA = 0.1                                             # train_v1.py:33 --sign-loss default


# d loss / d key cases (tools/gen_golden.py: dkey_cases runs them on the reference's own autograd)
DKEY_GEOMETRIES = {
    # name      ci  co  ks  s  pd  bk  hw  n  norm    relu
    'bk3_s2':  (8, 16, 3, 2, 1, 3, 9, 5, 'none', False),     # key batch 3 (mean over b, :152,173), stride 2
    'bn_s1':   (4, 16, 3, 1, 1, 1, 8, 6, 'bn', True),        # the layer4 form: BatchNorm, stride 1
    'sc_1x1':  (8, 16, 1, 2, 0, 2, 8, 4, 'bn', True),        # a projection shortcut: 1x1, stride 2, no padding
}


def dkey_inputs(name):
    """Deterministic inputs of one d/dkey case, shared with the tests (numpy legacy RandomState)."""
    ci, co, ks, s, pd, bk, hw, n, norm, relu = DKEY_GEOMETRIES[name]
    rs = np.random.RandomState(100 + sorted(DKEY_GEOMETRIES).index(name))
    ho = (hw + 2 * pd - ks) // s + 1
    return dict(
        w=(rs.standard_normal((co, ci, ks, ks)) * 0.2).astype(np.float32),
        key=rs.uniform(-1, 1, (bk, ci, hw, hw)).astype(np.float32),
        skey=rs.uniform(-1, 1, (bk, ci, hw, hw)).astype(np.float32),
        x=rs.standard_normal((n, ci, hw, hw)).astype(np.float32),
        cot=rs.standard_normal((n, co, ho, ho)).astype(np.float32),
        b=np.where(rs.uniform(size=co) < 0.5, -1.0, 1.0).astype(np.float32))
