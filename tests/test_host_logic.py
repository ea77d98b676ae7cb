"""CPU tests of the product's HOST logic (module API, autograd wiring, trainers, config surface).

The HIP kernels cannot run here, so `passport_ops.kernels` is monkeypatched with the oracle-backed
stand-in of tests/oracle_kernels.py; everything above the kernel boundary is the shipped code.  The
same cases run through the real kernels in tests/test_parity_gpu.py."""
import json
import os
import random

import numpy as np
import pytest
import torch

from oracle import runner
from oracle.cases import CASES
from tests.compare import close, compare_case
from tests.impls import ProductImpl, load_golden
from tests.oracle_kernels import OracleKernels

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cpu_kernels(monkeypatch):
    from deepipr_amd import passport_ops
    monkeypatch.setattr(passport_ops, 'kernels', OracleKernels())
    return passport_ops


def test_product_refuses_cpu_tensors():
    """No CPU fallback: the un-patched product raises on host tensors."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    blk = PassportBlock(4, 16, 3, 1, 1, {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1})
    with pytest.raises(RuntimeError, match='GPU only'):
        blk(torch.randn(2, 4, 8, 8))


@pytest.mark.parametrize('fuse_norm', [True, False])
@pytest.mark.parametrize('name', list(CASES))
def test_host_wiring_matches_reference(name, fuse_norm, golden_dir, cpu_kernels):
    if not fuse_norm and CASES[name]['norm'] == 'none':
        pytest.skip('no norm, nothing to fuse')
    torch.set_num_threads(8)
    gold = load_golden(golden_dir, name)
    got = runner.collect(name, ProductImpl('cpu', fuse_norm=fuse_norm))
    # InstanceNorm over 64-element planes amplifies fp32 rounding on low-variance planes (invstd up to 316): the
    # reference's own fp32 gradients carry ~1e-4 of noise there, which the fused path (float64 statistics) does not
    # reproduce; everything up to the losses still agrees to 2e-5
    noisy = ('grad/', 'post/', 'logits_eval/') if (fuse_norm and CASES[name]['norm'] == 'in') else ()
    compare_case(got, gold, rtol=2e-5, atol=2e-6, skip_prefixes=('train/acc',) + noisy)
    for k in gold:
        if k.startswith(noisy) and noisy:
            _close(got[k], gold[k], k, 5e-4, 2e-4)
    # same torch seed -> same constructor RNG draws as the reference: signature vectors identical
    for k in gold:
        if k.startswith('ctor_b/'):
            assert np.array_equal(got[k], gold[k]), k
    for k in ('train/acc', 'train/acc_public', 'train/acc_private'):
        if k in gold:
            assert got[k] == pytest.approx(float(gold[k]), abs=1e-4)


def test_passport_selection_and_set_key(golden_dir):
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    gold = load_golden(golden_dir, 'blocks')
    blk = PassportBlock(8, 16, 3, 2, 1, {'norm_type': 'none', 'key_type': 'random', 'sign_loss': 0.5}, relu=False)
    cands = torch.arange(5 * 6 * 2 * 2, dtype=torch.float32).view(5, 6, 2, 2)
    random.seed(1234)
    assert np.array_equal(blk.passport_selection(cands).numpy(), gold['selection/c6'])
    cands3 = torch.arange(5 * 3 * 2 * 2, dtype=torch.float32).view(5, 3, 2, 2)
    random.seed(1234)
    assert np.array_equal(blk.passport_selection(cands3).numpy(), gold['selection/c3'])
    random.seed(99)
    blk.set_key(cands, cands + 1000)
    assert np.array_equal(blk.key.numpy(), gold['selection/set_key_key'])
    assert np.array_equal(blk.skey.numpy(), gold['selection/set_key_skey'])


def test_block_level_bk3(golden_dir, cpu_kernels):
    """Key batch 3 (mean over b), stride 2, relu=False, alpha 0.5: y, gamma, beta, sign loss, dW, dx."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    gold = load_golden(golden_dir, 'blocks')
    rs = np.random.RandomState(7)
    w = torch.from_numpy(rs.standard_normal((16, 8, 3, 3)).astype(np.float32) * 0.2)
    key = torch.from_numpy(rs.uniform(-1, 1, (3, 8, 9, 9)).astype(np.float32))
    skey = torch.from_numpy(rs.uniform(-1, 1, (3, 8, 9, 9)).astype(np.float32))
    x = torch.from_numpy(rs.standard_normal((5, 8, 9, 9)).astype(np.float32)).requires_grad_(True)
    cot = torch.from_numpy(rs.standard_normal((5, 16, 5, 5)).astype(np.float32))
    b = torch.from_numpy(np.where(rs.uniform(size=16) < 0.5, -1.0, 1.0).astype(np.float32))
    blk = PassportBlock(8, 16, 3, 2, 1, {'norm_type': 'none', 'key_type': 'random', 'sign_loss': 0.5}, relu=False)
    with torch.no_grad():
        blk.weight.copy_(w)
        blk.b.copy_(b)
        blk.sign_loss.b.copy_(b)
    blk.register_buffer('key', key)
    blk.register_buffer('skey', skey)
    y = blk(x)
    ((y * cot).sum() + blk.sign_loss.loss).backward()
    from tests.compare import close
    close(y.detach().numpy(), gold['bk3/y'], 'y', 2e-5, 2e-6)
    close(blk.sign_loss.scale_cache.detach().numpy().reshape(-1), gold['bk3/gamma'], 'gamma', 2e-5, 2e-6)
    close(blk.get_bias().detach().numpy().reshape(-1), gold['bk3/beta'], 'beta', 2e-5, 2e-6)
    close(float(blk.sign_loss.loss), gold['bk3/sign_loss'], 'loss', 2e-5, 2e-6)
    close(blk.weight.grad.numpy(), gold['bk3/dW'], 'dW', 1e-4, 1e-5)
    close(x.grad.numpy(), gold['bk3/dx'], 'dx', 1e-4, 1e-5)


def test_trainable_keys_get_gradients(cpu_kernels):
    """passport_attack_3.py:232-243 turns the keys into nn.Parameters: d/dkey must flow."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    torch.manual_seed(3)
    blk = PassportBlock(4, 16, 3, 1, 1, {'norm_type': 'none', 'key_type': 'random', 'sign_loss': 0.1})
    blk.key = torch.nn.Parameter(torch.rand(1, 4, 6, 6) * 2 - 1)
    blk.skey = torch.nn.Parameter(torch.rand(1, 4, 6, 6) * 2 - 1)
    x = torch.randn(3, 4, 6, 6)
    y = blk(x)
    (y.square().sum() + blk.sign_loss.loss).backward()
    ref_w = blk.weight.detach().clone().requires_grad_(True)
    ks, kb = blk.skey.detach().clone().requires_grad_(True), blk.key.detach().clone().requires_grad_(True)
    conv = lambda t: torch.nn.functional.conv2d(t, ref_w, None, 1, 1)
    g = conv(ks).mean(dim=(0, 2, 3))
    bt = conv(kb).mean(dim=(0, 2, 3))
    yy = torch.relu(g.view(1, -1, 1, 1) * conv(x) + bt.view(1, -1, 1, 1))
    sl = (0.1 * torch.relu(-blk.b * g + 0.1)).sum() + 1e-5 * g.pow(2).sum()
    (yy.square().sum() + sl).backward()
    assert torch.allclose(blk.skey.grad, ks.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(blk.key.grad, kb.grad, rtol=1e-4, atol=1e-6)
    assert torch.allclose(blk.weight.grad, ref_w.grad, rtol=1e-4, atol=1e-5)


def test_state_dict_roundtrip_presizes_lazy_tensors(cpu_kernels):
    """_load_from_state_dict pre-sizes key/skey/scale/bias so a strict load works on a fresh model
    (passportconv2d.py:177-196)."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    for cls, names in ((PassportBlock, ('key', 'skey')), (PassportPrivateBlock, ('key_private', 'skey_private'))):
        a = cls(4, 16, 3, 1, 1, kw)
        a(torch.randn(2, 4, 8, 8))
        sd = a.state_dict()
        for n in names:
            assert n in sd and tuple(sd[n].shape) == (1, 4, 8, 8)
        b = cls(4, 16, 3, 1, 1, kw)
        b.load_state_dict(sd, strict=True)
        for k in sd:
            assert torch.equal(b.state_dict()[k], sd[k]), k
    a = PassportBlock(4, 16, 3, 1, 1, kw)
    a.init_scale(True)
    a.init_bias(True)
    b = PassportBlock(4, 16, 3, 1, 1, kw)
    b.load_state_dict(a.state_dict(), strict=True)
    assert isinstance(b.scale, torch.nn.Parameter) and isinstance(b.bias, torch.nn.Parameter)


def test_passport_kwargs_builder_matches_reference_shape():
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    cfg = json.load(open(os.path.join(ROOT, 'passport_configs', 'resnet18_passport.json')))
    kw, keys = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'shuffle',
                                                    'sl_ratio': 0.1}, need_index=True)
    assert keys == ['layer4.0.convbnrelu_1', 'layer4.0.convbn_2', 'layer4.0.shortcut', 'layer4.1.convbnrelu_1',
                    'layer4.1.convbn_2']
    assert kw['layer4']['0']['shortcut'] == {'flag': True, 'norm_type': 'bn', 'key_type': 'shuffle', 'sign_loss': 0.1}
    assert kw['convbnrelu_1']['flag'] is False
    cfg = json.load(open(os.path.join(ROOT, 'passport_configs', 'alexnet_passport.json')))
    cfg['4'] = 'abc'
    kw, keys = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'gn', 'key_type': 'random',
                                                    'sl_ratio': 0.5}, need_index=True)
    assert keys == ['4', '5', '6'] and kw['4']['b'] == 'abc' and kw['4']['flag'] is True


def _ckpt(gold, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in gold.items() if k.startswith(prefix)}


def test_reference_checkpoints_load_strictly_and_reproduce_outputs(golden_dir, cpu_kernels):
    """state_dicts written by the REFERENCE's PassportBlock / PassportPrivateBlock load with
    strict=True into fresh blocks of this build (lazily created key buffers included) and give the reference's
    evaluation outputs."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
    from tests.compare import close
    gold = load_golden(golden_dir, 'blocks')
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    x = torch.from_numpy(gold['ckpt_in/x'])
    sd = _ckpt(gold, 'ckpt_v1/')
    assert set(sd) == {'weight', 'conv.weight', 'b', 'sign_loss.b', 'key', 'skey', 'bn.running_mean',
                       'bn.running_var', 'bn.num_batches_tracked'}
    for fuse in (True, False):
        blk = PassportBlock(4, 16, 3, 1, 1, kw)
        blk.fuse_norm = fuse
        blk.load_state_dict(sd, strict=True)
        blk.eval()
        with torch.no_grad():
            close(blk(x).numpy(), gold['ckpt_v1_out/y'], 'v1 eval output', 2e-5, 2e-6)
    sd = _ckpt(gold, 'ckpt_private/')
    assert {'scale', 'bias', 'key_private', 'skey_private', 'sign_loss_private.b'} <= set(sd)
    blk = PassportPrivateBlock(4, 16, 3, 1, 1, kw)
    blk.load_state_dict(sd, strict=True)
    blk.eval()
    with torch.no_grad():
        close(blk(x, ind=0).numpy(), gold['ckpt_private_out/y0'], 'public eval output', 2e-5, 2e-6)
        close(blk(x, ind=1).numpy(), gold['ckpt_private_out/y1'], 'private eval output', 2e-5, 2e-6)
    # and the other direction: a state_dict written here has exactly the reference's key set
    mine = PassportBlock(4, 16, 3, 1, 1, kw)
    mine(x)
    assert set(mine.state_dict()) == set(_ckpt(gold, 'ckpt_v1/'))


def test_resnet50_bottleneck_passport_matches_composed_oracle(cpu_kernels):
    """BASELINE config 5 (ResNet50 passport) has no reference implementation; the build's Bottleneck variant is
    pinned against the oracle's blocks (each pinned to the reference) composed the same way.  CPU, tiny input."""
    from oracle import patterns, torch_ref
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.resnet_passport import ResNet50Passport
    cfg = json.load(open(os.path.join(ROOT, 'passport_configs', 'resnet50_passport.json')))
    kw, keys = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random',
                                                    'sl_ratio': 0.1}, need_index=True)
    assert len(keys) == 10 and keys[0] == 'layer4.0.convbnrelu_1' and 'layer4.0.shortcut' in keys
    torch.manual_seed(0)
    np.random.seed(0)
    prod = ResNet50Passport(num_classes=10, passport_kwargs=kw)
    ref = torch_ref.resnet50_ref(num_classes=10, passport_kwargs=torch_ref.passport_kwargs_from_config(
        cfg, 'bn', 'random', 0.1))
    assert sorted(prod.state_dict().keys() - {k for k in prod.state_dict() if k.endswith('key')}) == \
        sorted(ref.state_dict().keys())
    x, y = patterns.batch(16, 3, 32, 32, 10)
    prod.train(), ref.train()
    with torch.no_grad():
        prod(x), ref(x)
    patterns.fill_state(prod), patterns.fill_state(ref)
    out_p, out_r = prod(x), ref(x)
    assert torch.allclose(out_p, out_r, rtol=1e-4, atol=1e-5), float((out_p - out_r).abs().max())
    sp = sum(m.sign_loss.loss for m in prod.modules() if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
    sr = sum(m.loss for m in torch_ref.sign_losses(ref))
    assert float(sp) == pytest.approx(float(sr), rel=1e-5)
    (torch.nn.functional.cross_entropy(out_p, y) + sp).backward()
    (torch.nn.functional.cross_entropy(out_r, y) + sr).backward()
    gp = dict(prod.named_parameters())
    for name, p in ref.named_parameters():
        scale = float(p.grad.abs().max()) + 1e-12
        # 50 layers of batch-norm over a batch of 4 amplify rounding differences on the way back to the stem;
        # layer4 (the passport layers) and the classifier are the meaningful comparison
        tol = 1e-2 if name.startswith(('layer4', 'linear')) else 5e-2
        assert float((gp[name].grad - p.grad).abs().max()) <= tol * scale + 1e-7, name


def test_flat_sgd_matches_torch_sgd_and_round_trips_state(cpu_kernels):
    """FlatSGD on the host (oracle-backed SGD kernel): same trajectory as torch.optim.SGD, parameters live in one
    flat buffer, lr schedulers act on it, state_dict carries the momentum."""
    from deepipr_amd.flat_sgd import FlatSGD
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net_b = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    net_b.load_state_dict(net_a.state_dict())
    oa = torch.optim.SGD(net_a.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    ob = FlatSGD(net_b.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, [2], 0.1)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, [2], 0.1)
    assert all(p.data_ptr() >= ob.flat_param.data_ptr() for p in net_b.parameters())
    x = torch.randn(9, 7)
    for i in range(4):
        for net, opt, sch in ((net_a, oa, sa), (net_b, ob, sb)):
            opt.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
            opt.step()
            sch.step()
        if i == 1:                                                   # checkpoint the optimiser mid-run
            saved = ob.state_dict()
            ob.load_state_dict(saved)
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7)
    assert ob.param_groups[0]['lr'] == pytest.approx(0.01)
    assert 'flat_momentum' in ob.state_dict()


def test_sign_loss_module_api(cpu_kernels):
    """SignLoss.add / get_loss (hinge only) / get_acc / set_b / reset / the 'scale_cache is None' errors
    (models/losses/sign_loss.py:15-59)."""
    from deepipr_amd.models.losses.sign_loss import SignLoss
    from oracle import np_passport as npp
    rs = np.random.RandomState(4)
    g = torch.from_numpy((rs.standard_normal(48) * 0.2).astype(np.float32))
    b = torch.from_numpy(np.where(rs.uniform(size=48) < 0.5, -1.0, 1.0).astype(np.float32))
    sl = SignLoss(0.3, b.clone())
    with pytest.raises(Exception, match='scale_cache is None'):
        sl.get_loss()
    with pytest.raises(Exception, match='scale_cache is None'):
        sl.get_acc()
    sl.add(g.view(1, -1, 1, 1))
    sl.add(g.view(1, -1, 1, 1))                                       # accumulates (loss += ..., acc += ...)
    loss64, acc64, _ = npp.sign_loss_fwd(g.numpy().astype(np.float64), b.numpy().astype(np.float64), 0.3)
    assert float(sl.loss) == pytest.approx(2 * float(loss64), rel=1e-5)
    assert float(sl.acc) == pytest.approx(2 * float(acc64), rel=1e-6)
    hinge = (0.3 * np.maximum(-b.numpy() * g.numpy() + 0.1, 0)).sum()
    assert float(sl.get_loss()) == pytest.approx(float(hinge), rel=1e-5)   # no L2 term here (sign_loss.py:25-30)
    assert float(sl.get_acc()) == pytest.approx(float(acc64), rel=1e-6)
    sl.set_b(-b)
    assert float(sl.get_acc()) == pytest.approx(1.0 - float(acc64), abs=1e-6)
    sl.reset()
    assert sl.loss == 0 and sl.acc == 0 and sl.scale_cache is None
    assert 'b' in sl.state_dict()


def test_imagenet_geometries_and_mixed_flags(cpu_kernels):
    """ImageNet-shaped stems/classifiers (alexnet_passport.py:29-30,65-82; resnet_passport.py:94-98), a passport
    stem, and a block with convbnrelu_1 passported but convbn_2 plain (the reference's BasicPassportBlock
    mis-dispatches that combination, resnet_passport.py:72; here it simply works)."""
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.alexnet_passport import AlexNetPassport
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    from deepipr_amd.models.resnet_passport import ResNet18Passport, ResNet9Passport
    from oracle.cases import alexnet_config, resnet18_config
    mk = lambda cfg: construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn',
                                                          'key_type': 'random', 'sl_ratio': 0.1})
    torch.manual_seed(0)
    a = AlexNetPassport(3, 1000, mk(alexnet_config()))
    assert a.features[0].conv.kernel_size == (11, 11) and len(a.classifier) == 7
    a.eval()
    assert tuple(a(torch.randn(1, 3, 224, 224)).shape) == (1, 1000)
    cfg = resnet18_config()
    cfg['convbnrelu_1'] = True                                       # passport stem
    cfg['layer1']['0'] = {'convbnrelu_1': True, 'convbn_2': False}   # mixed block
    r = ResNet18Passport(num_classes=10, passport_kwargs=mk(cfg), imagenet=True)
    assert isinstance(r.convbnrelu_1[0], PassportBlock) and r.convbnrelu_1[0].conv.kernel_size == (7, 7)
    r.train()
    out = r(torch.randn(2, 3, 64, 64))
    assert tuple(out.shape) == (2, 10)
    sd = r.state_dict()
    assert tuple(sd['convbnrelu_1.0.key'].shape) == (1, 3, 64, 64) and 'layer1.0.convbnrelu_1.skey' in sd
    cfg9 = json.load(open(os.path.join(ROOT, 'passport_configs', 'resnet9_passport.json')))
    r9 = ResNet9Passport(num_classes=10, passport_kwargs=mk(cfg9))
    assert sum(isinstance(m, PassportBlock) for m in r9.modules()) == 3


def _check_shuttle(got, want, tol=0.0):
    for k in sorted(got):
        assert k in want, 'scenario output %s has no golden' % k
        if tol:
            np.testing.assert_allclose(got[k], want[k], rtol=tol, atol=tol, err_msg=k)
        else:
            assert np.array_equal(got[k], want[k]), k


def test_weight_shuttles_match_reference(golden_dir):
    """experiments/utils.py:100-239 -- plain->passport (V1 and private), passport->plain through the learnable
    pair, plain->plain with a different class count; pure tensor copies, so every digest is bit-identical."""
    from tests.impls import ProductShuttle
    want = load_golden(golden_dir, 'shuttle')
    got = runner.collect_shuttle(ProductShuttle('cpu'), with_keys=False)
    assert len(got) > 600
    _check_shuttle(got, want)


def _close(*a):
    from tests.compare import close
    return close(*a)


def test_force_passport_paths(golden_dir, cpu_kernels):
    """flip_attack.py:25 / pruning_attack.py:26 / passportconv2d.py:142-175: with the learnable pair installed
    (init_scale/init_bias(True), what the weight shuttles do) plain calls use it, force_passport=True returns to the
    key-derived gamma/beta and refreshes the sign loss; private blocks: force_passport overrides ind=0."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
    gold = load_golden(golden_dir, 'blocks')
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    x = torch.from_numpy(gold['ckpt_in/x']).to('cpu')
    for fuse in (True, False):
        blk = PassportBlock(4, 16, 3, 1, 1, kw)
        blk.fuse_norm = fuse
        blk.load_state_dict(_ckpt(gold, 'ckpt_v1/'), strict=True)
        blk = blk.to('cpu').eval()
        blk.init_scale(True)
        blk.init_bias(True)
        assert blk.scale.device == x.device and blk.scale.requires_grad
        with torch.no_grad():
            blk.scale.copy_(torch.from_numpy(gold['force/scale']))
            blk.bias.copy_(torch.from_numpy(gold['force/bias']))
            _close(blk(x).cpu().numpy(), gold['force/v1_plain'], 'learnable pair', 2e-5, 2e-6)
            assert np.array_equal(blk.get_scale().cpu().numpy().reshape(-1), gold['force/scale'])
            blk.sign_loss.reset()
            _close(blk(x, force_passport=True).cpu().numpy(), gold['force/v1_forced'], 'forced', 2e-5, 2e-6)
            g = blk.get_scale(True).cpu().numpy().reshape(-1)
            _close(g, gold['force/v1_forced_scale'], 'forced gamma', 1e-5, 1e-6)
            assert np.array_equal(np.sign(g), np.sign(gold['force/v1_forced_scale']))
            _close(blk.get_bias(True).cpu().numpy().reshape(-1), gold['force/v1_forced_bias'], 'forced beta', 1e-5, 1e-6)
            _close(np.float64(float(blk.sign_loss.loss)), gold['force/v1_forced_sign_loss'], 'sign loss', 1e-5, 1e-6)
            assert float(blk.sign_loss.acc) == float(gold['force/v1_forced_sign_acc'])
    pv = PassportPrivateBlock(4, 16, 3, 1, 1, kw)
    pv.load_state_dict(_ckpt(gold, 'ckpt_private/'), strict=True)
    pv = pv.to('cpu').eval()
    with torch.no_grad():
        _close(pv(x, force_passport=True, ind=0).cpu().numpy(), gold['force/private_forced_ind0'],
              'private forced', 2e-5, 2e-6)


@pytest.mark.parametrize('name', ['bk3_s2', 'bn_s1', 'sc_1x1'])
def test_trainable_keys_match_reference_autograd(name, golden_dir, cpu_kernels):
    """Keys as nn.Parameters (passport_attack_3.py:232-243): the product's autograd wiring of d/dkey, d/dskey and the
    three-way dW against the REFERENCE's own autograd (goldens blocks.npz: dkey/*).  GPU twin: test_passport_layer_gpu.py."""
    from tests.blocks import run_dkey_case
    gold = load_golden(golden_dir, 'blocks')
    got = run_dkey_case(name, 'cpu')
    for k, v in got.items():
        close(v, gold['dkey/%s/%s' % (name, k)], k, 1e-4, 1e-5)


def test_signloss_b_and_alpha_drive_the_fused_loss(cpu_kernels):
    """passport_attack_3.py:261: m.sign_loss.set_b(newb) must change the training loss (models/losses/sign_loss.py:27
    uses SignLoss.b / SignLoss.alpha), also on the fused norm paths which compute the loss inside the layer kernel."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    torch.manual_seed(4)
    np.random.seed(4)
    x = torch.randn(4, 4, 8, 8)
    for norm in ('bn', 'gn', 'none'):
        blk = PassportBlock(4, 32, 3, 1, 1, {'norm_type': norm, 'key_type': 'random', 'sign_loss': 0.5})
        blk(x)
        gamma = blk.sign_loss.scale_cache.detach().view(-1)
        before = float(blk.sign_loss.loss)
        newb = -torch.sign(gamma)                              # every bit wrong: the hinge is active everywhere
        blk.sign_loss.set_b(newb)
        blk.sign_loss.alpha = 0.25
        blk(x)
        want = float((0.25 * torch.relu(-newb * gamma + 0.1)).sum() + 1e-5 * gamma.pow(2).sum())
        assert float(blk.sign_loss.loss) == pytest.approx(want, rel=1e-5), norm
        assert float(blk.sign_loss.loss) != pytest.approx(before, rel=1e-3)
        assert float(blk.sign_loss.acc) == 0.0


def test_pooled_key_cache_is_dropped_when_keys_are_replaced(cpu_kernels):
    """set_key twice with no forward in between, a `.data` write and a load_state_dict must all be seen by the next
    forward (the cache of pooled passport means is keyed on address + version only as a fast path)."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    torch.manual_seed(5)
    kw = {'norm_type': 'bn', 'key_type': 'shuffle', 'sign_loss': 0.1}
    blk = PassportBlock(4, 16, 3, 1, 1, kw).eval()
    k1, k2 = torch.rand(1, 4, 8, 8) * 2 - 1, torch.rand(1, 4, 8, 8) * 2 - 1

    def gamma_direct(skey):
        return torch.nn.functional.conv2d(skey, blk.weight, None, 1, 1).mean(dim=(0, 2, 3)).detach()
    with torch.no_grad():
        blk.set_key(k1.clone(), k1.clone())
        assert torch.allclose(blk.get_scale().view(-1), gamma_direct(k1), rtol=1e-5, atol=1e-6)
        blk.set_key(k2.clone(), k2.clone())
        assert torch.allclose(blk.get_scale().view(-1), gamma_direct(k2), rtol=1e-5, atol=1e-6)
        blk.skey.data.copy_(k1)                                 # no version bump on `skey`
        blk.invalidate_key_cache()
        assert torch.allclose(blk.get_scale().view(-1), gamma_direct(k1), rtol=1e-5, atol=1e-6)
        sd = blk.state_dict()
        sd['skey'] = k2.clone()
        blk.load_state_dict(sd)
        assert torch.allclose(blk.get_scale().view(-1), gamma_direct(k2), rtol=1e-5, atol=1e-6)


def test_flat_sgd_skips_frozen_and_gradless_parameters_like_torch_sgd(cpu_kernels):
    """torch.optim.SGD (experiments/classification.py:47-50) leaves parameters with requires_grad=False or grad None
    untouched -- no weight decay, no momentum.  FlatSGD must do the same."""
    from deepipr_amd.flat_sgd import FlatSGD
    torch.manual_seed(6)

    def make():
        torch.manual_seed(6)
        net = torch.nn.ModuleList([torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)])
        net[1].weight.requires_grad_(False)                    # frozen
        return net
    a, b = make(), make()
    oa = torch.optim.SGD([p for p in a.parameters() if p.requires_grad], lr=0.1, momentum=0.9, weight_decay=1e-2)
    ob = FlatSGD(b.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-2)
    x = torch.randn(5, 8)
    for step in range(3):
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad(set_to_none=True)
            h = net[1](net[0](x))
            if step != 1:                                      # step 1: the last layer is unused -> its grads are None
                h = net[2](h)
            h.square().sum().backward()
            opt.step()
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-6), na
    assert torch.equal(a[1].weight, make()[1].weight)          # the frozen one never moved


def test_flat_sgd_device_hypers_follow_the_scheduler(cpu_kernels):
    """The update reads lr from FlatSGD's 4-float tensor; sync_hyper() tracks param_groups (MultiStepLR)."""
    from deepipr_amd.flat_sgd import FlatSGD
    lin = torch.nn.Linear(4, 4)
    opt = FlatSGD(lin.parameters(), lr=0.5, momentum=0.0, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [1], 0.1)
    w0 = lin.weight.detach().clone()
    lin.weight.grad = torch.ones_like(lin.weight)
    lin.bias.grad = torch.zeros_like(lin.bias)
    opt.step()
    assert torch.allclose(lin.weight, w0 - 0.5)
    sched.step()
    opt.step()
    assert torch.allclose(lin.weight, w0 - 0.5 - 0.05)
    assert opt._hyper.tolist() == pytest.approx([0.05, 0.0, 0.0, 1.0])


def test_conv_inside_node_with_frozen_weight_and_input(cpu_kernels):
    """The fused conv + passport node when neither the input nor the weight needs a gradient (a frozen first layer, the
    fine-tuning set-ups of the reference's transfer-learning harness): backward must not call convolution_backward
    with an all-false mask, gradients still reach trainable keys."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    torch.manual_seed(8)
    np.random.seed(8)
    blk = PassportBlock(4, 16, 3, 1, 1, {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1})
    x = torch.randn(3, 4, 6, 6)
    blk(x)                                                  # draws the keys
    blk.weight.requires_grad_(False)
    key = blk.key.clone()
    del blk.key
    blk.register_parameter('key', torch.nn.Parameter(key))
    y = blk(x)
    (y.sum() + blk.sign_loss.loss).backward()
    assert blk.weight.grad is None and blk.key.grad is not None and torch.isfinite(blk.key.grad).all()


@pytest.mark.parametrize('private', [False, True])
def test_staged_backward_equals_one_backward_pass(private, cpu_kernels):
    """experiments/staged.py: the backward cut into stages at the model's named activations (torch.autograd.grad on
    detached cut leaves, the objective kept as a root) must give EXACTLY the gradients and the step of one
    loss.backward().  ResNet18 with passports -- hence sign losses -- in layer3 AND layer4: a sign loss of a layer in
    a later stage reaches its layer only through the objective root, not through a cut; the residual blocks hand
    their outputs out as two handles, so every cut is a pair; V2 marks every cut twice (public + private forward)."""
    from deepipr_amd.experiments.staged import StagedStep
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.flat_sgd import FlatSGD
    from deepipr_amd.models.resnet_passport import ResNet18Passport
    from deepipr_amd.models.resnet_passport_private import ResNet18Private
    from oracle.cases import resnet18_config
    torch.set_num_threads(8)
    cfg = resnet18_config()
    cfg['layer3'] = {b: {k: True for k in v} for b, v in cfg['layer3'].items()}
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random',
                                              'sl_ratio': 0.1})

    def make():
        torch.manual_seed(11)
        np.random.seed(11)
        net = (ResNet18Private if private else ResNet18Passport)(num_classes=10, passport_kwargs=kw)
        net.train()
        with torch.no_grad():
            net(torch.randn(2, 3, 32, 32))
        return net
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
    step = train_step_v23 if private else train_step_v1
    a, b = make(), make()
    wa, wb = (DualBranch(a), DualBranch(b)) if private else (a, b)
    oa = FlatSGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)     # the same update arithmetic on both sides
    ob = FlatSGD(b.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    staged = StagedStep(step, wb, ob, x, y, graph=False, warmup=0)
    assert [s.cut for s in staged.stages] == ['layer4.0', 'layer3.0', None] and ob._mode == 'staged'
    assert sum(len(s.params) for s in staged.stages) == len(list(b.parameters()))
    for it in range(2):
        out_a = step(wa, oa, x, y)
        grads_a = {k: p.grad.clone() for k, p in a.named_parameters()}
        out_b = staged(x, y)
        for k, p in b.named_parameters():
            assert torch.equal(p.grad, grads_a[k]), (it, k, float((p.grad - grads_a[k]).abs().max()))
        for u, v in zip(out_a, out_b):
            assert torch.equal(u, v)
    for (k, u), (_, v) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(u, v), k


@pytest.mark.parametrize('arch', ['resnet18', 'alexnet'])
def test_shared_trunk_of_the_dual_forward_equals_two_full_passes(arch, cpu_kernels, monkeypatch):
    """V2 / V3 step (trainer_private.py:159-171: model(x, ind=0) and model(x, ind=1) over the same batch): the layers in
    front of the first private passport layer run ONCE for both branches (models/_builders.shared_trunk).  Against the
    two full passes (DEEPIPR_NO_SHARED_TRUNK=1): identical logits, gradients equal up to the association of the two
    branches' sum, and -- the one thing the reference does twice that has an effect -- the batch-norm running
    statistics after two updates with the same batch statistic, num_batches_tracked advanced by two; the net's own
    forward hooks still see two calls."""
    from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from oracle.cases import alexnet_config, resnet18_config
    torch.set_num_threads(8)
    if arch == 'resnet18':
        from deepipr_amd.models.resnet_passport_private import ResNet18Private
        kw = construct_passport_kwargs_from_dict({'passport_config': resnet18_config(), 'norm_type': 'bn',
                                                  'key_type': 'random', 'sl_ratio': 0.1})
        ctor = lambda: ResNet18Private(num_classes=10, passport_kwargs=kw)
        shared_bn, branch_bn = 'layer3.1.convbn_2.bn', 'layer4.0.convbnrelu_1.bn'
    else:
        from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
        kw = construct_passport_kwargs_from_dict({'passport_config': alexnet_config(), 'norm_type': 'bn',
                                                  'key_type': 'random', 'sl_ratio': 0.1})
        ctor = lambda: AlexNetPassportPrivate(3, 10, kw)
        shared_bn, branch_bn = 'features.2.bn', 'features.4.bn'

    def make():
        torch.manual_seed(5)
        np.random.seed(5)
        net = ctor()
        net.train()
        with torch.no_grad():
            net(torch.randn(2, 3, 32, 32))
        return net
    g = torch.Generator().manual_seed(9)
    x, y = torch.randn(6, 3, 32, 32, generator=g), torch.randint(0, 10, (6,), generator=g)
    res = {}
    for mode in ('shared', 'twice'):
        if mode == 'twice':
            monkeypatch.setenv('DEEPIPR_NO_SHARED_TRUNK', '1')
        else:
            monkeypatch.delenv('DEEPIPR_NO_SHARED_TRUNK', raising=False)
        net = make()
        seen = []
        net.register_forward_hook(lambda _m, _i, o: seen.append(o.detach().clone()))
        opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
        calls = {'n': 0}
        stem_conv = net.convbnrelu_1.conv if arch == 'resnet18' else net.features[0].conv
        # counted by wrapping .forward: a hook on a submodule makes DualBranch take the two full passes, and
        # a module hook would make the layer keep its convolution to itself
        stem_inner = stem_conv.forward
        stem_conv.forward = lambda inp, _f=stem_inner: (calls.__setitem__('n', calls['n'] + 1), _f(inp))[1]
        # the first private layer behind the split: its data convolution is shared by the two branches as well
        first = (net.layer4[0].convbnrelu_1 if arch == 'resnet18' else net.features[4]).conv
        inner = first.forward
        first.forward = lambda inp, _f=inner: (calls.__setitem__('first', calls.get('first', 0) + 1), _f(inp))[1]
        train_step_v23(DualBranch(net), opt, x, y)
        res[mode] = dict(logits=seen, grads={k: p.grad.clone() for k, p in net.named_parameters()},
                         state={k: v.clone() for k, v in net.state_dict().items()}, stem_calls=calls['n'],
                         first_calls=calls.get('first', 0))
    a, b = res['shared'], res['twice']
    assert (a['stem_calls'], b['stem_calls']) == (1, 2)                  # the trunk really ran once
    assert (a['first_calls'], b['first_calls']) == (1, 2)               # and so did the first convolution behind the split
    assert len(a['logits']) == len(b['logits']) == 2                    # the net's forward hooks: two calls
    for u, v in zip(a['logits'], b['logits']):
        assert torch.equal(u, v)
    for k in b['grads']:
        scale = float(b['grads'][k].abs().max()) + 1e-12
        # one backward pass over the trunk with the branches' summed gradient against the sum of two passes: fp32
        # reassociation through up to 13 layers, a few 1e-6 of scale (the north star's bar is 1e-4)
        assert float((a['grads'][k] - b['grads'][k]).abs().max()) <= 2e-5 * scale + 1e-9, k
    for k, v in b['state'].items():
        if k.endswith('num_batches_tracked'):
            assert int(a['state'][k]) == int(v) == 3, k          # the key-drawing forward in make() + two passes
        else:
            assert torch.allclose(a['state'][k], v, rtol=1e-5, atol=1e-7), k
    # and the statistics did move twice: one update would leave running_mean at 0.1 * batch mean, two at 0.19 *
    for name in (shared_bn, branch_bn):
        assert float(a['state'][name + '.running_mean'].abs().max()) > 0


def test_alexnet_split_at_the_stage_cut_does_not_share_the_convolution(cpu_kernels):
    """AlexNet V2 whose FIRST private passport layer is features[5] -- exactly where backward_stages() cuts the staged
    backward (alexnet_passport._CUT).  The cut is marked on that layer's input, per branch; a convolution shared in front
    of the mark would tie the two stages' autograd graphs together, so forward_dual keeps one convolution per branch there
    (the trunk in front of it is still shared).  Equal to the two full passes."""
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models import alexnet_passport
    from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
    from oracle.cases import alexnet_config
    cfg = dict(alexnet_config())
    cfg['4'] = False                                        # passport layers: features 5 and 6 only
    assert cfg['5'] and cfg['6'] and alexnet_passport._CUT == 5
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random',
                                              'sl_ratio': 0.1})
    torch.manual_seed(3)
    np.random.seed(3)
    net = AlexNetPassportPrivate(3, 10, kw)
    net.train()
    x = torch.randn(4, 3, 32, 32)
    with torch.no_grad():
        net(x)
    calls = {'first': 0, 'stem': 0}
    first = net.features[5].conv
    inner = first.forward
    first.forward = lambda inp, _f=inner: (calls.__setitem__('first', calls['first'] + 1), _f(inp))[1]
    stem = net.features[0].conv
    inner0 = stem.forward
    stem.forward = lambda inp, _f=inner0: (calls.__setitem__('stem', calls['stem'] + 1), _f(inp))[1]
    state = {k: v.clone() for k, v in net.state_dict().items()}
    a0, a1 = net.forward_dual(x)
    assert calls == {'first': 2, 'stem': 1}                 # trunk shared, the convolution at the cut is not
    net.load_state_dict(state)
    b0, b1 = net(x, ind=0), net(x, ind=1)
    assert torch.equal(a0, b0) and torch.equal(a1, b1)


def test_stage_groups_follow_layers_swapped_in_after_the_first_lookup(cpu_kernels):
    """the layer -> backward-stage table is keyed on the module objects; a block replaced after the first
    forward (fine-tune / attack scripts) is filed under its real stage, not -1, and nothing is kept in the model."""
    import copy
    from deepipr_amd import passport_ops as P
    from deepipr_amd.models.resnet_passport import ResNet18Passport
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from oracle.cases import resnet18_config
    torch.manual_seed(0)
    kw = construct_passport_kwargs_from_dict({'passport_config': resnet18_config(), 'norm_type': 'bn',
                                              'key_type': 'random', 'sl_ratio': 0.1})
    net = ResNet18Passport(num_classes=10, passport_kwargs=kw)
    stages = net.backward_stages()
    group_of = P.stage_groups(net)
    last = stages[-1][1][-1]
    inner = next(m for m in last.modules() if m is not last)
    k = group_of(inner)
    assert k == len(stages) - 1 and group_of(torch.nn.ReLU()) == -1
    # swap a submodule of that stage for a fresh object
    name, old = next((n, m) for n, m in last.named_children())
    fresh = copy.deepcopy(old)
    setattr(last, name, fresh)
    assert P.stage_groups(net)(fresh) == k
    assert '_stage_of' not in net.__dict__
    clone = copy.deepcopy(net)
    assert P.stage_groups(clone)(next(m for m in clone.backward_stages()[0][1][0].modules())) == 0


def test_dual_branch_takes_two_full_passes_when_a_submodule_carries_a_hook(cpu_kernels):
    """hooks the one-pass form would fire once (or never) force model(x, ind=0); model(x, ind=1)."""
    from deepipr_amd.experiments.trainer_private import DualBranch, _submodule_hooks

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.trunk = torch.nn.Linear(3, 3)
            self.calls = []
        def forward(self, x, ind=0):
            self.calls.append(('forward', ind))
            return self.trunk(x) + ind
        def forward_dual(self, x):
            self.calls.append(('dual',))
            t = self.trunk(x)
            return t, t + 1

    net, x = Net(), torch.zeros(2, 3)
    DualBranch(net)(x)
    assert net.calls == [('dual',)] and not _submodule_hooks(net)
    seen = []
    handle = net.trunk.register_forward_hook(lambda m, i, o: seen.append(1))
    net.calls.clear()
    DualBranch(net)(x)
    assert net.calls == [('forward', 0), ('forward', 1)] and len(seen) == 2
    handle.remove()
    net.register_full_backward_hook(lambda m, gi, go: None)
    net.calls.clear()
    DualBranch(net)(x)
    assert net.calls == [('forward', 0), ('forward', 1)]
