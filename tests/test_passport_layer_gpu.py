"""The passport layer (models/layers/passportconv2d.py:142-223) on the GPU: d/dkey against the reference's own autograd,
signature bits on adversarial rows, one layer's whole backward chain at config-R shapes, the batched GEMV and the
(grouped / fused) rank-2 weight-gradient term, SignLoss.set_b.  Every call goes through the C ABI -> HIP kernels."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import patterns, torch_ref
from oracle.cases import ALPHA, SGD, alexnet_config
from tests.compare import close, states_close
from tests.gpu_common import DEV, K, dev, host, pinned_miopen      # noqa: F401  (K is a fixture)
from tests.impls import load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------- d/dkey vs the reference's autograd
@pytest.mark.parametrize('name', ['bk3_s2', 'bn_s1', 'sc_1x1'])
def test_trainable_keys_match_reference_autograd_on_gpu(K, name, golden_dir):
    """deepipr_gamma_beta_dkey (+ the layer's whole backward) against gradients produced by the REFERENCE's own
    autograd with key / skey turned into nn.Parameters (passport_attack_3.py:232-243)."""
    from tests.blocks import run_dkey_case
    gold = load_golden(golden_dir, 'blocks')
    got = run_dkey_case(name, DEV)
    for k, v in got.items():
        close(v, gold['dkey/%s/%s' % (name, k)], k, 1e-4, 1e-5)


# ----------------------------------------------------------------------------- adversarial signature rows
def test_signature_bits_on_adversarial_near_zero_rows(K, golden_dir):
    """Rows of W tuned so that the exact gamma is +-1e-2 ... +-1e-8 (the smallest far below the fp32 summation noise
    of the 144-term dot product), evaluated by the REFERENCE's own get_scale (goldens blocks.npz: nearzero/*,
    passportconv2d.py:142-158).  The kernel accumulates in f64 and rounds once, so
      * its sign(gamma) is the sign of the exact sum on EVERY row, and
      * it equals the reference's sign and value wherever the reference's own fp32 answer is numerically meaningful
        (|gamma| above 8 eps * sum|W_k m_k|); below that bound the reference itself flips 12 of the 96 signs relative
        to the exact sum, which no implementation can or should reproduce."""
    from oracle import np_passport as npp
    gold = load_golden(golden_dir, 'blocks')
    w, skey, key, g_ref = (gold['nearzero/' + k] for k in ('w', 'skey', 'key', 'gamma_ref'))
    co = w.shape[0]
    s, n = npp.pooled_patch_sum(skey.astype(np.float64), 3, 3, 1, 1)
    exact = w.reshape(co, -1).astype(np.float64) @ (s / n)
    bound = 8 * 6e-8 * (np.abs(w.reshape(co, -1).astype(np.float64)) * np.abs(s / n)).sum(axis=1)
    m = K.pooled_patch_mean(dev(np.stack([skey, key])), 3, 3, 1, 1)
    gamma = host(K.gamma_beta_fwd(dev(w), m)[0])
    assert np.array_equal(np.sign(gamma), np.sign(exact)), 'sign of the exact sum, every row'
    assert np.all(np.abs(gamma - exact) <= 1.2e-7 * np.abs(exact) + 1e-30)           # one rounding of the exact value
    meaningful = np.abs(exact) >= bound
    assert meaningful.sum() >= 40 and (~meaningful).sum() >= 20                     # the fixture spans both regimes
    assert np.array_equal(np.sign(gamma[meaningful]), np.sign(g_ref[meaningful]))
    assert np.all(np.abs(gamma - g_ref)[meaningful] <= bound[meaningful])
    # and the layer API reads the same bits out (TesterPrivate.test_signature path)
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    blk = PassportBlock(w.shape[1], co, 3, 1, 1, {'norm_type': 'none', 'key_type': 'random', 'sign_loss': 0.1})
    with torch.no_grad():
        blk.weight.copy_(torch.from_numpy(w))
    blk = blk.to(DEV)
    blk.set_key(dev(key), dev(skey))
    with torch.no_grad():
        bits = host(blk.get_scale().view(-1).sign())
    assert np.array_equal(bits, np.sign(exact))


# ----------------------------------------------------------------------------- one layer, whole backward chain
@pytest.mark.miopen_pinned
@pytest.mark.parametrize('fuse_norm', [True, False])
@pytest.mark.parametrize('geom', [(256, 512, 3, 2, 1, 8), (512, 512, 3, 1, 1, 4), (256, 512, 1, 2, 0, 8)])
def test_passport_block_backward_chain_at_config_R_shape(K, geom, fuse_norm):
    """One PassportBlock (conv -> BatchNorm -> passport affine -> ReLU, + sign loss) at the shapes of config R's
    layer4 (batch 128): y, dx and the THREE-WAY dW (data conv wgrad + gamma + beta contributions,
    models/layers/passportconv2d.py:148,169,218) against the same layer evaluated in float64 on the host.
    ReLU kinks are taken out of the comparison explicitly: the cotangent is zeroed on every element whose float64
    pre-activation is within 1e-4 of zero (counted and bounded), so a mask that differs there cannot matter and
    everything else has to agree to 1e-4 of its scale."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    ci, co, ks, s, pd, hw = geom
    n = 128
    rs = np.random.RandomState(ci + co + ks)
    x = rs.standard_normal((n, ci, hw, hw)).astype(np.float32)
    wt = (rs.standard_normal((co, ci, ks, ks)) * np.sqrt(2.0 / (co * ks * ks))).astype(np.float32)
    key = rs.uniform(-1, 1, (1, ci, hw, hw)).astype(np.float32)
    skey = rs.uniform(-1, 1, (1, ci, hw, hw)).astype(np.float32)
    b = np.where(rs.uniform(size=co) < 0.5, -1.0, 1.0).astype(np.float32)
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': ALPHA}
    # float64 reference on the host (stock ATen ops of oracle/torch_ref.py)
    ref = torch_ref.PassportLayerRef(ci, co, ks, s, pd, kw)
    with torch.no_grad():
        ref.weight.copy_(torch.from_numpy(wt))
        ref.b.copy_(torch.from_numpy(b))               # shared with ref.sign_loss.b until .double() splits them
    ref = ref.double()
    assert torch.equal(ref.b, ref.sign_loss.b)
    ref.set_key(torch.from_numpy(key).double(), torch.from_numpy(skey).double())
    ref.train()
    xr = torch.from_numpy(x).double().requires_grad_(True)
    xc = ref.conv(xr)
    z = ref.get_scale() * ref.bn(xc) + ref.get_bias()
    yr = torch.relu(z)
    cot = rs.standard_normal(tuple(yr.shape))
    near = (z.detach().abs() < 1e-4).numpy()
    cot[near] = 0.0
    assert near.mean() < 1e-2, near.sum()                   # gamma ~ 0.05: ~0.15 % of the pre-activations
    (yr * torch.from_numpy(cot)).sum().add(ref.sign_loss.loss).backward()
    # product on the GPU
    blk = PassportBlock(ci, co, ks, s, pd, kw)
    blk.fuse_norm = fuse_norm
    with torch.no_grad():
        blk.weight.copy_(torch.from_numpy(wt))
        blk.b.copy_(torch.from_numpy(b))
        blk.sign_loss.b.copy_(torch.from_numpy(b))
    blk = blk.to(DEV).train()
    blk.set_key(dev(key), dev(skey))
    xg = dev(x).requires_grad_(True)
    with pinned_miopen():
        yg = blk(xg)
        ((yg * dev(cot)).sum() + blk.sign_loss.loss).backward()
        torch.cuda.synchronize()
    y_ref = yr.detach().numpy()
    flips = ((host(yg) > 0) != (y_ref > 0)) & ~near
    assert flips.sum() == 0, 'ReLU masks may only differ within 1e-4 of the kink'
    ok = ~near
    assert np.abs(host(yg) - y_ref)[ok].max() <= 1e-4 * max(1.0, np.abs(y_ref).max())
    close(host(blk.sign_loss.scale_cache).reshape(-1), ref.sign_loss.scale_cache.detach().numpy().reshape(-1),
          'gamma', 1e-5, 1e-6)
    assert abs(float(blk.sign_loss.loss) - float(ref.sign_loss.loss)) <= 1e-5 * max(1.0, float(ref.sign_loss.loss))
    for name, got, want in (('dx', xg.grad, xr.grad), ('dW (three-way)', blk.weight.grad, ref.weight.grad)):
        want = want.numpy()
        scale = np.abs(want).max()
        err = np.abs(host(got) - want).max()
        assert err <= 1e-4 * scale, (name, err, scale)
    close(host(blk.bn.running_var), ref.bn.running_var.numpy(), 'running_var', 1e-5, 1e-6)


# ----------------------------------------------------------------------------- accumulate-into dW
@pytest.mark.parametrize('co,kk', [(512, 4608), (512, 256), (384, 1728), (64, 75), (5, 7)])
def test_gamma_beta_bwd_accumulates_into_an_existing_wgrad(K, co, kk):
    rs = np.random.RandomState(co + kk)
    m = dev(rs.uniform(-1, 1, (2, kk)), torch.float64)
    dg, db = dev(rs.standard_normal(co)), dev(rs.standard_normal(co))
    base = dev(rs.standard_normal((co, kk)))
    fresh = K.gamma_beta_bwd(dg, db, m, (co, kk))
    acc = K.gamma_beta_bwd_acc(dg, db, m, base.clone())
    assert torch.equal(acc, base + fresh)                      # one rounding of the same sum


# ----------------------------------------------------------------------------- SignLoss.set_b on the fused path
def test_set_b_changes_the_fused_training_loss(K):
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    torch.manual_seed(4)
    np.random.seed(4)
    x = torch.randn(16, 8, 8, 8, device=DEV)
    for norm in ('bn', 'gn', 'none'):
        blk = PassportBlock(8, 32, 3, 1, 1, {'norm_type': norm, 'key_type': 'random', 'sign_loss': 0.5}).to(DEV)
        blk(x)
        gamma = blk.sign_loss.scale_cache.detach().view(-1)
        newb = -torch.sign(gamma)
        blk.sign_loss.set_b(newb)                              # passport_attack_3.py:261
        blk.sign_loss.alpha = 0.25
        blk(x)
        want = float((0.25 * torch.relu(-newb * gamma + 0.1)).sum() + 1e-5 * gamma.pow(2).sum())
        assert float(blk.sign_loss.loss.detach()) == pytest.approx(want, rel=1e-5), norm
        assert float(blk.sign_loss.acc) == 0.0


# ----------------------------------------------------------------------------- batched GEMV pair
GEMV_BATCHES = [
    [(512, 2304), (512, 4608), (512, 256), (512, 4608), (512, 4608)],     # ResNet18 layer4 (config R / P)
    [(384, 1728), (256, 3456), (256, 2304)],                              # AlexNet features 4-6 (config A)
    [(5, 7), (64, 75), (33, 1028), (2, 5124), (1, 4)],                    # K % 4 != 0, K beyond one trip, odd row counts
    [(512, 4608)],                                                        # n = 1 (what the single-layer entry points call)
]


@pytest.mark.parametrize('batch', GEMV_BATCHES)
def test_gamma_beta_multi_matches_the_oracle_and_the_single_layer_calls(K, batch):
    """deepipr_gamma_beta_fwd_multi / _bwd_multi: gamma, beta of every layer of a batch in one launch, and the rank-2
    update of all their dW in one launch (fresh and accumulate form) -- against the float64 contraction W . m that
    oracle/np_passport.py: gamma_beta_fwd reduces to for pooled keys (gamma within 1 ulp-ish of the f64 sum: 2e-7 relative to sum |W m|; signature bits exact) and bit-identical to the per-layer
    calls (the same kernel with n = 1)."""
    rs = np.random.RandomState(len(batch) * 1000 + batch[0][1])
    ws = [dev(rs.standard_normal((co, k)) * 0.05) for co, k in batch]
    ms = [dev(rs.uniform(-1, 1, (2, k)), torch.float64) for co, k in batch]
    got = K.gamma_beta_fwd_multi(ws, ms)
    for (g, b), w, m in zip(got, ws, ms):
        w64, m64 = host(w).astype(np.float64), host(m)
        g64, b64 = w64 @ m64[0], w64 @ m64[1]
        mag = np.abs(w64) @ np.abs(m64[0]) + 1e-30
        assert np.all(np.abs(host(g) - g64) <= 2e-7 * mag + 1e-30)
        assert np.all(np.abs(host(b) - b64) <= 2e-7 * (np.abs(w64) @ np.abs(m64[1])) + 1e-30)
        meaningful = np.abs(g64) > 8 * np.finfo(np.float32).eps * mag
        assert np.array_equal(np.sign(host(g))[meaningful], np.sign(g64)[meaningful])       # signature bits
        g1, b1 = K.gamma_beta_fwd(w, m)
        assert torch.equal(g1, g) and torch.equal(b1, b)
    dgs = [dev(rs.standard_normal(co)) for co, k in batch]
    dbs = [dev(rs.standard_normal(co)) for co, k in batch]
    fresh = K.gamma_beta_bwd_multi(dgs, dbs, ms, [torch.full_like(w, float('nan')) for w in ws], False)
    for dw, dg, db, m, w in zip(fresh, dgs, dbs, ms, ws):
        m32 = host(m).astype(np.float32)
        ref = (host(dg)[:, None].astype(np.float64) * m32[0][None, :] + host(db)[:, None].astype(np.float64) * m32[1][None, :])
        assert np.abs(host(dw) - ref).max() <= 2e-6 * (np.abs(ref).max() + 1e-30)
        assert torch.equal(dw, K.gamma_beta_bwd(dg, db, m, tuple(w.shape)))
    base = [dev(rs.standard_normal(tuple(w.shape))) for w in ws]
    acc = K.gamma_beta_bwd_multi(dgs, dbs, ms, [t.clone() for t in base], True)
    for a, t, f in zip(acc, base, fresh):
        assert torch.equal(a, t + f)                                  # one rounding of the same sum


# ----------------------------------------------------------------------------- rank-2 update of a layer group in one launch
@pytest.mark.parametrize('own_wgrad', [True, False])
@pytest.mark.parametrize('net_kind', ['resnet18_v1', 'resnet18_v2', 'resnet18_v2_private_pass_first', 'alexnet_v1'])
def test_grouped_rank2_update_is_bit_identical_to_the_per_layer_updates(K, net_kind, own_wgrad, monkeypatch):
    """The passport branch's dW of a group of layers in ONE launch (_Rank2Group, the default) against one launch per
    layer (DEEPIPR_NO_RANK2_BATCH=1) and against no batching at all (DEEPIPR_NO_GEMV_BATCH=1): logits and every
    parameter gradient bit-identical with MIOpen pinned; the launch counts say which form ran.  The AlexNet's passport
    layers span two backward stages (features 4 | 5, 6): the groups follow the stages."""
    from deepipr_amd import _lib
    from deepipr_amd import passport_ops as P
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from oracle.cases import alexnet_config, resnet18_config
    monkeypatch.setattr(P, 'OWN_WGRAD', own_wgrad)            # False: the vendor library's wgrad + the separate rank-2 update
    private = 'v2' in net_kind
    rev = net_kind.endswith('first')     # the private pass's nodes are then the OLDER ones: the other backward order
    if net_kind.startswith('resnet18'):
        from deepipr_amd.models.resnet_passport import ResNet18Passport
        from deepipr_amd.models.resnet_passport_private import ResNet18Private
        cfg, ctor, n = resnet18_config(), (ResNet18Private if private else ResNet18Passport), 32
    else:
        from deepipr_amd.models.alexnet_passport import AlexNetPassport
        cfg, ctor, n = alexnet_config(), AlexNetPassport, 64
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random',
                                              'sl_ratio': 0.1})
    torch.manual_seed(3)
    np.random.seed(3)
    if net_kind.startswith('resnet18'):
        net = ctor(num_classes=10, passport_kwargs=kw).to(DEV)
    else:
        net = ctor(3, 10, kw).to(DEV)
    net.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (n,), generator=g).to(DEV)
    with torch.no_grad():
        net(x)                                                # keys
    state = {k: v.clone() for k, v in net.state_dict().items()}
    ce = torch.nn.functional.cross_entropy

    def step(env):
        for k in ('DEEPIPR_NO_RANK2_BATCH', 'DEEPIPR_NO_GEMV_BATCH'):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, '1')
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        _lib.profile_enable(True)
        if private:
            outs = [net(x, ind=1), net(x, ind=0)] if rev else [net(x, ind=0), net(x, ind=1)]
            loss = ce(outs[0], y) + ce(outs[1], y) + sum(m.sign_loss_private.loss for m in net.modules()
                                                         if hasattr(m, 'sign_loss_private'))
        else:
            outs = [net(x)]
            loss = ce(outs[0], y) + sum(m.sign_loss.loss for m in net.modules()
                                        if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
        loss.backward()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        prof = _lib.profile_read()
        got = {'logits%d' % i: o.detach().clone() for i, o in enumerate(outs)}
        got.update({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        return got, int(prof['gamma_beta_fwd'][1]), int(prof['gamma_beta_bwd'][1])

    with pinned_miopen():
        grouped, f0, b0 = step(())
        per_layer, f1, b1 = step(('DEEPIPR_NO_RANK2_BATCH',))
        unbatched, f2, b2 = step(('DEEPIPR_NO_GEMV_BATCH',))
    n_layers = 3 if net_kind == 'alexnet_v1' else 5
    fwd_passes = 1                                              # V2: only the private pass (ind = 1) uses the passports
    # layers whose weight gradient deepipr_conv_wgrad computes get their rank-2 term in ITS reduction pass (no launch of
    # their own, grouped or not) -- since the 1x1 stride-2 instance all passport layers of both nets; a layer outside the
    # kernel (other geometry, DEEPIPR_OWN_WGRAD=0) keeps the separate update, one launch per group or per layer
    from deepipr_amd.models._builders import PASSPORT_TYPES
    shape = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp: shape.__setitem__(mod, inp[0].shape))
             for m in net.modules() if isinstance(m, PASSPORT_TYPES)]
    with torch.no_grad():
        net(x, ind=1) if private else net(x)
    for h in hooks:
        h.remove()
    net.load_state_dict(state)
    separate = [m for m, shp in shape.items()
                if not P._own_wgrad(torch.empty(shp, device=DEV), m.weight, m.conv.stride[0], m.conv.padding[0])]
    assert len(shape) == n_layers and len(separate) == (0 if own_wgrad else n_layers)
    assert (f1, b1) == (fwd_passes, len(separate) * fwd_passes), (f1, b1)
    assert (f2, b2) == (n_layers * fwd_passes, len(separate) * fwd_passes), (f2, b2)
    # one launch per group that still has such layers: ResNet18's five layers are one stage; AlexNet: features 5, 6
    # together, features 4 alone
    groups = 0 if own_wgrad else (2 if net_kind == 'alexnet_v1' else 1)
    assert (f0, b0) == (fwd_passes, groups), (f0, b0)
    for name, other in (('per layer', per_layer), ('unbatched', unbatched)):
        assert set(other) == set(grouped)
        diff = {k: float((grouped[k] - other[k]).abs().max()) for k in grouped if not torch.equal(grouped[k], other[k])}
        assert not diff, '%s: %d tensors differ: %s' % (name, len(diff), sorted(diff.items(), key=lambda kv: -kv[1])[:4])
