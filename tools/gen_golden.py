#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference (kamwoh/DeepIPR) on CPU.

Build-container only: imports /root/reference (read-only), which does not exist on the GPU box.
Only *outputs* (logits, gamma/beta, sign bits, losses, gradient digests) are written; weights and
inputs come from oracle.patterns' name-keyed deterministic fills, so no reference source or
pickled module ever enters this repo.

    python tools/gen_golden.py            # all cases
    python tools/gen_golden.py resnet18_v1
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('DEEPIPR_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)

# torchvision is not installed; the reference only *calls* these for ImageNet-pretrained nets
# (models/alexnet_passport.py:3,84-102; models/resnet_passport.py:4,123-135).
_tv = types.ModuleType('torchvision')
_tvm = types.ModuleType('torchvision.models')
_tvm.alexnet = _tvm.resnet18 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('no torchvision'))
_tv.models = _tvm
sys.modules.setdefault('torchvision', _tv)
sys.modules.setdefault('torchvision.models', _tvm)
sys.path.insert(0, REF)

import passport_generator                                            # noqa: E402
from experiments.trainer import Trainer                              # noqa: E402
from experiments.trainer_private import TesterPrivate, TrainerPrivate  # noqa: E402
from experiments.utils import (construct_passport_kwargs_from_dict, load_normal_model_to_normal_model,  # noqa: E402
                               load_normal_model_to_passport_model, load_passport_model_to_normal_model)
from models.alexnet_normal import AlexNetNormal                      # noqa: E402
from models.alexnet_passport import AlexNetPassport                  # noqa: E402
from models.alexnet_passport_private import AlexNetPassportPrivate   # noqa: E402
from models.layers.passportconv2d import PassportBlock               # noqa: E402
from models.layers.passportconv2d_private import PassportPrivateBlock  # noqa: E402
from models.resnet_normal import ResNet9, ResNet18                   # noqa: E402
from models.resnet_passport import ResNet9Passport, ResNet18Passport  # noqa: E402
from models.resnet_passport_private import ResNet18Private           # noqa: E402

from oracle import runner                                            # noqa: E402
from oracle.cases import (ALPHA, CASES, DKEY_GEOMETRIES, alexnet_config, dkey_inputs,   # noqa: E402
                          resnet18_config)


class ReferenceImpl:
    device = torch.device('cpu')

    def build(self, case):
        kw = construct_passport_kwargs_from_dict({'passport_config': case['config'],
                                                  'norm_type': case['norm'],
                                                  'key_type': case.get('key_type', 'random'), 'sl_ratio': ALPHA})
        private = case['scheme'] != 1
        if case['arch'] == 'alexnet':
            cls = AlexNetPassportPrivate if private else AlexNetPassport
            return cls(3, case['ncls'], kw)
        if case['arch'] == 'resnet9':
            assert not private
            return ResNet9Passport(num_classes=case['ncls'], passport_kwargs=kw)
        cls = ResNet18Private if private else ResNet18Passport
        return cls(num_classes=case['ncls'], passport_kwargs=kw)

    def plain(self, case):
        """The key-propagation net of --key-type shuffle (experiments/classification.py:68-100)."""
        if case['arch'] == 'alexnet':
            return AlexNetNormal(3, case['ncls'], case['norm'])
        if case['arch'] == 'resnet9':
            return ResNet9(num_classes=case['ncls'], norm_type=case['norm'])
        return ResNet18(num_classes=case['ncls'], norm_type=case['norm'])

    def set_keys(self, plain, model, kx, ky):
        passport_generator.set_key(plain, model, kx, ky)

    def is_passport(self, m):
        return isinstance(m, (PassportBlock, PassportPrivateBlock))

    def is_private(self, m):
        return isinstance(m, PassportPrivateBlock)

    def step(self, model, opt, batch, wm):
        private = any(isinstance(m, PassportPrivateBlock) for m in model.modules())
        tr = (TrainerPrivate if private else Trainer)(model, opt, None, self.device)
        return tr.train(0, [batch], [wm] if wm is not None else None)

    def test_signature(self, model):
        return TesterPrivate(model, self.device, verbose=False).test_signature()


class ReferenceShuttle:
    """The reference's weight shuttles (experiments/utils.py:100-239) on its own nets, CPU."""
    device = torch.device('cpu')
    n2p = staticmethod(load_normal_model_to_passport_model)
    n2n = staticmethod(load_normal_model_to_normal_model)
    p2n = staticmethod(load_passport_model_to_normal_model)

    def _kw(self, arch):
        cfg = alexnet_config() if arch == 'alexnet' else resnet18_config()
        return construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn',
                                                    'key_type': 'random', 'sl_ratio': ALPHA}, True)

    def plkeys(self, arch):
        return self._kw(arch)[1]

    def plain(self, arch, ncls):
        return AlexNetNormal(3, ncls, 'bn') if arch == 'alexnet' else ResNet18(num_classes=ncls, norm_type='bn')

    def passport(self, arch, ncls, private):
        kw = self._kw(arch)[0]
        if arch == 'alexnet':
            return (AlexNetPassportPrivate if private else AlexNetPassport)(3, ncls, kw)
        return (ResNet18Private if private else ResNet18Passport)(num_classes=ncls, passport_kwargs=kw)


def block_cases():
    """Layer-level vectors the model cases do not reach: key batch > 1 (mean over b,
    passportconv2d.py:152,173), relu=False, stride 2; and passport_selection (:90-123)."""
    out = {}
    torch.manual_seed(0)
    blk = PassportBlock(8, 16, 3, 2, 1, {'norm_type': 'none', 'key_type': 'random', 'sign_loss': 0.5}, relu=False)
    rs = np.random.RandomState(7)
    w = torch.from_numpy(rs.standard_normal((16, 8, 3, 3)).astype(np.float32) * 0.2)
    key = torch.from_numpy(rs.uniform(-1, 1, (3, 8, 9, 9)).astype(np.float32))
    skey = torch.from_numpy(rs.uniform(-1, 1, (3, 8, 9, 9)).astype(np.float32))
    x = torch.from_numpy(rs.standard_normal((5, 8, 9, 9)).astype(np.float32))
    cot = torch.from_numpy(rs.standard_normal((5, 16, 5, 5)).astype(np.float32))
    b = torch.from_numpy(np.where(rs.uniform(size=16) < 0.5, -1.0, 1.0).astype(np.float32))
    with torch.no_grad():
        blk.weight.copy_(w)
        blk.b.copy_(b)
    blk.register_buffer('key', key)          # bypass set_key's n>1 selection on purpose
    blk.register_buffer('skey', skey)
    x.requires_grad_(True)
    y = blk(x)
    total = (y * cot).sum() + blk.sign_loss.loss
    total.backward()
    out['bk3/y'] = y.detach().numpy()
    out['bk3/gamma'] = blk.sign_loss.scale_cache.detach().numpy().reshape(-1)
    out['bk3/beta'] = blk.get_bias().detach().numpy().reshape(-1)
    out['bk3/sign_loss'] = np.float64(blk.sign_loss.loss.item())
    out['bk3/sign_acc'] = np.float64(float(blk.sign_loss.acc))
    out['bk3/dW'] = blk.weight.grad.numpy().copy()
    out['bk3/dx'] = x.grad.numpy().copy()

    # checkpoint compatibility: state_dicts written by the reference's blocks after one training
    # forward (lazily created keys included), plus their evaluation-mode outputs on a fixed input
    rs = np.random.RandomState(11)
    xin = torch.from_numpy(rs.standard_normal((6, 4, 8, 8)).astype(np.float32))
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    torch.manual_seed(5)
    np.random.seed(5)
    v1 = PassportBlock(4, 16, 3, 1, 1, kw)
    v1.train()
    v1(xin)
    v1.eval()
    for k, v in v1.state_dict().items():
        out['ckpt_v1/' + k] = v.numpy().copy()
    with torch.no_grad():
        out['ckpt_v1_out/y'] = v1(xin).numpy()
    pv = PassportPrivateBlock(4, 16, 3, 1, 1, kw)
    pv.train()
    pv(xin, ind=0)
    pv(xin, ind=1)
    with torch.no_grad():
        pv.scale.add_(0.3 * torch.randn(16))
        pv.bias.add_(0.3 * torch.randn(16))
    pv.eval()
    for k, v in pv.state_dict().items():
        out['ckpt_private/' + k] = v.numpy().copy()
    with torch.no_grad():
        out['ckpt_private_out/y0'] = pv(xin, ind=0).numpy()
        out['ckpt_private_out/y1'] = pv(xin, ind=1).numpy()
    out['ckpt_in/x'] = xin.numpy()

    # force_passport (flip_attack.py:25, pruning_attack.py:26, passportconv2d.py:142-175): once a V1 block carries
    # the learnable pair (init_scale/init_bias(True), as the weight shuttles do), plain calls use it and
    # force_passport=True goes back to the key-derived gamma/beta and refreshes the sign loss
    rs = np.random.RandomState(13)
    out['force/scale'] = (1.0 + 0.5 * rs.standard_normal(16)).astype(np.float32)
    out['force/bias'] = (0.5 * rs.standard_normal(16)).astype(np.float32)
    v1.init_scale(True)
    v1.init_bias(True)
    with torch.no_grad():
        v1.scale.copy_(torch.from_numpy(out['force/scale']))
        v1.bias.copy_(torch.from_numpy(out['force/bias']))
        out['force/v1_plain'] = v1(xin).numpy()
        out['force/v1_plain_scale'] = v1.get_scale().numpy().reshape(-1)
        v1.sign_loss.reset()
        out['force/v1_forced'] = v1(xin, force_passport=True).numpy()
        out['force/v1_forced_scale'] = v1.get_scale(True).numpy().reshape(-1)
        out['force/v1_forced_bias'] = v1.get_bias(True).numpy().reshape(-1)
        out['force/v1_forced_sign_loss'] = np.float64(v1.sign_loss.loss.item())
        out['force/v1_forced_sign_acc'] = np.float64(float(v1.sign_loss.acc))
        out['force/private_forced_ind0'] = pv(xin, force_passport=True, ind=0).numpy()

    # passport_selection is driven by python's `random` module
    cands = torch.arange(5 * 6 * 2 * 2, dtype=torch.float32).view(5, 6, 2, 2)
    random.seed(1234)
    out['selection/c6'] = blk.passport_selection(cands).numpy()
    cands3 = torch.arange(5 * 3 * 2 * 2, dtype=torch.float32).view(5, 3, 2, 2)
    random.seed(1234)
    out['selection/c3'] = blk.passport_selection(cands3).numpy()
    # set_key with n>1 goes through the selection for both tensors (:128-131)
    random.seed(99)
    blk.set_key(cands, cands + 1000)
    out['selection/set_key_key'] = blk.key.numpy().copy()
    out['selection/set_key_skey'] = blk.skey.numpy().copy()
    out.update(dkey_cases())
    out.update(nearzero_case())
    return out


def nearzero_case():
    """Adversarial signature rows: weights tuned so that the EXACT gamma of each output channel is a prescribed tiny
    value (down to 1e-8, far below fp32 summation noise of a 144-term dot product), then evaluated by the reference's
    own get_scale (fp32 oneDNN conv + means, passportconv2d.py:142-158).  The fixture carries the inputs (small) and
    the reference's gamma; the tests check that the kernels' sign(gamma) equals the reference's wherever the
    reference's own answer is numerically meaningful, and the exact sign everywhere."""
    from oracle import np_passport as npp
    rs = np.random.RandomState(21)
    co, ci, hw = 96, 16, 6
    w = (rs.standard_normal((co, ci, 3, 3)) * 0.2).astype(np.float32)
    skey = rs.uniform(-1, 1, (1, ci, hw, hw)).astype(np.float32)
    key = rs.uniform(-1, 1, (1, ci, hw, hw)).astype(np.float32)
    s, n = npp.pooled_patch_sum(skey.astype(np.float64), 3, 3, 1, 1)
    m = s / n                                                   # [K] f64, K = ci*9
    k0 = 4                                                      # centre tap of input channel 0
    mags = [1e-2, 1e-4, 1e-5, 1e-6, 3e-7, 1e-7, 3e-8, 1e-8]
    wf = w.reshape(co, -1).astype(np.float64)
    for c in range(co):
        target = mags[c % len(mags)] * (1.0 if (c // len(mags)) % 2 == 0 else -1.0)
        g = wf[c] @ m
        wf[c, k0] -= (g - target) / m[k0]
    w = wf.astype(np.float32).reshape(co, ci, 3, 3)
    torch.manual_seed(0)
    blk = PassportBlock(ci, co, 3, 1, 1, {'norm_type': 'none', 'key_type': 'random', 'sign_loss': 0.1})
    with torch.no_grad():
        blk.weight.copy_(torch.from_numpy(w))
    blk.set_key(torch.from_numpy(key), torch.from_numpy(skey))
    with torch.no_grad():
        gamma = blk.get_scale().view(-1).numpy().copy()
    return {'nearzero/w': w, 'nearzero/skey': skey, 'nearzero/key': key, 'nearzero/gamma_ref': gamma}


def dkey_cases():
    """d loss / d key and d skey from the reference's OWN autograd, keys turned into nn.Parameters the way
    passport_attack_3.py:232-243 does (delattr the buffer, register_parameter under the same name)."""
    out = {}
    for name, (ci, co, ks, s, pd, bk, hw, n, norm, relu) in DKEY_GEOMETRIES.items():
        t = {k: torch.from_numpy(v) for k, v in dkey_inputs(name).items()}
        torch.manual_seed(0)
        blk = PassportBlock(ci, co, ks, s, pd, {'norm_type': norm, 'key_type': 'random', 'sign_loss': 0.5},
                            relu=relu)
        with torch.no_grad():
            blk.weight.copy_(t['w'])
            blk.b.copy_(t['b'])
        blk.__delattr__('key')
        blk.__delattr__('skey')
        blk.register_parameter('key', torch.nn.Parameter(t['key'].clone()))
        blk.register_parameter('skey', torch.nn.Parameter(t['skey'].clone()))
        blk.train()
        x = t['x'].clone().requires_grad_(True)
        y = blk(x)
        ((y * t['cot']).sum() + blk.sign_loss.loss).backward()
        pre = 'dkey/' + name + '/'
        out[pre + 'y'] = y.detach().numpy()
        out[pre + 'gamma'] = blk.sign_loss.scale_cache.detach().numpy().reshape(-1)
        out[pre + 'sign_loss'] = np.float64(blk.sign_loss.loss.item())
        out[pre + 'dkey'] = blk.key.grad.numpy().copy()
        out[pre + 'dskey'] = blk.skey.grad.numpy().copy()
        out[pre + 'dW'] = blk.weight.grad.numpy().copy()
        out[pre + 'dx'] = x.grad.numpy().copy()
    return out


def main(argv):
    names = argv or (list(CASES) + ['blocks', 'shuttle'])
    os.makedirs(os.path.join(ROOT, 'tests', 'golden'), exist_ok=True)
    torch.set_num_threads(8)
    for name in names:
        if name == 'blocks':
            out = block_cases()
        elif name == 'shuttle':
            out = runner.collect_shuttle(ReferenceShuttle(), with_keys=True)
        else:
            out = runner.collect(name, ReferenceImpl())
        path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
        np.savez_compressed(path, **{k.replace('/', '|'): v for k, v in out.items()})
        print('%-18s %4d arrays  %7.1f KB' % (name, len(out), os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main(sys.argv[1:])
