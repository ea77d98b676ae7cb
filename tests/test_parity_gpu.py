"""GPU parity tests: every call goes through the C ABI (ctypes -> libdeepipr_hip.so -> HIP kernels).

Checker = oracle/ (numpy + plain PyTorch on the host) and tests/golden/*.npz (outputs of the real
reference).  Tolerances: fp32 results within 1e-4 (north-star bar; most ops are much tighter and say
so), signature bits and ReLU masks exact.
"""
import os

import numpy as np
import pytest
import torch

from oracle import np_passport as npp
from oracle import patterns, runner, torch_ref
from oracle.cases import ALPHA, CASES, SGD, resnet18_config
from tests.compare import close, states_close, compare_case
from tests.impls import ProductImpl, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def K():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from deepipr_amd import _lib, passport_ops
    _lib.lib()                                   # fail loudly if the HIP library is missing
    assert type(passport_ops.kernels).__name__ == 'HipKernels'
    return passport_ops.kernels


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- affine (+ReLU)
AFFINE_SHAPES = [
    (128, 512, 4, 4),     # config R, every layer4 passport output (4.19 MB)
    (64, 384, 8, 8),      # config A, features.4
    (32, 512, 4, 4),      # config P per-GPU shard
    (6, 64, 32, 32),      # stem-sized plane: 1024 floats = 256 float4 (one workgroup row)
    (3, 16, 56, 56),      # plane larger than a workgroup (large-plane kernel)
    (5, 24, 7, 7),        # 7x7: plane not a multiple of 4 floats (scalar path)
    (2, 3, 5, 3),         # tiny, odd everything
    (1, 1, 1, 1),         # degenerate
    (7, 10, 2, 2),        # partial channel tiles
    (130, 20, 1, 1),      # HW = 1 (fully-connected style)
    (16, 64, 7, 7),       # ImageNet layer4 maps: float4 units straddling channels (C % 4 == 0)
    (4, 8, 9, 9),         # 9x9 planes, 81 floats
    (3, 12, 15, 15),      # 225-float planes, 4-channel tiles
]


@pytest.mark.parametrize('shape', AFFINE_SHAPES)
@pytest.mark.parametrize('relu', [True, False])
def test_affine_relu_fwd_bwd(K, shape, relu):
    rs = np.random.RandomState(sum(shape) + int(relu))
    n, c, h, w = shape
    x = rs.standard_normal(shape).astype(np.float32)
    g = rs.standard_normal(c).astype(np.float32)
    b = rs.standard_normal(c).astype(np.float32)
    if c > 2:
        g[1] = 0.0                                # gamma = 0: mask depends on beta alone
        b[2] = 0.0
    dy = rs.standard_normal(shape).astype(np.float32)
    y = host(K.affine_relu_fwd(dev(x), dev(g), dev(b), relu))
    want = npp.affine_relu_fwd(x, g, b, relu)     # f32 mul, f32 add: same two roundings
    assert np.array_equal(y, want), 'forward must be bit-exact (max|d|=%g)' % np.abs(y - want).max()

    dx, dg, db = [host(t) for t in K.affine_relu_bwd(dev(dy), dev(x), dev(g), dev(b), relu)]
    mask = (want > 0) if relu else np.ones_like(want, dtype=bool)
    dz = np.where(mask, dy, np.float32(0))
    assert np.array_equal(dx, dz * g.reshape(1, -1, 1, 1)), 'dxhat must be bit-exact'
    dg64 = (dz.astype(np.float64) * x.astype(np.float64)).sum(axis=(0, 2, 3))
    db64 = dz.astype(np.float64).sum(axis=(0, 2, 3))
    mag = np.abs(dz.astype(np.float64) * x).sum(axis=(0, 2, 3)) + 1e-30
    assert np.all(np.abs(dg - dg64) <= 4e-7 * mag + 1e-30), np.abs(dg - dg64).max()
    magb = np.abs(dz.astype(np.float64)).sum(axis=(0, 2, 3)) + 1e-30
    assert np.all(np.abs(db - db64) <= 4e-7 * magb + 1e-30), np.abs(db - db64).max()


def test_affine_bwd_is_deterministic(K):
    """Fixed-order reductions: two launches give bit-identical gradients."""
    rs = np.random.RandomState(5)
    x, dy = dev(rs.standard_normal((128, 512, 4, 4))), dev(rs.standard_normal((128, 512, 4, 4)))
    g, b = dev(rs.standard_normal(512)), dev(rs.standard_normal(512))
    a = [host(t).copy() for t in K.affine_relu_bwd(dy, x, g, b, True)]
    for _ in range(3):
        again = [host(t) for t in K.affine_relu_bwd(dy, x, g, b, True)]
        for u, v in zip(a, again):
            assert np.array_equal(u, v)


def test_affine_linearity_at_full_size(K):
    """Size-independent property at config-R size: without ReLU the layer is linear in (gamma, beta):
    f(x; g1+g2, b1+b2) == f(x; g1, b1) + f(x; g2, b2) up to fp32 rounding of the final add."""
    rs = np.random.RandomState(11)
    x = dev(rs.standard_normal((128, 512, 4, 4)))
    g1, g2, b1, b2 = [dev(rs.standard_normal(512)) for _ in range(4)]
    lhs = K.affine_relu_fwd(x, g1 + g2, b1 + b2, False)
    rhs = K.affine_relu_fwd(x, g1, b1, False) + K.affine_relu_fwd(x, g2, b2, False)
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-5)
    # and ReLU output is idempotent under a second identity-affine ReLU pass
    y = K.affine_relu_fwd(x, g1, b1, True)
    one, zero = torch.ones(512, device=DEV), torch.zeros(512, device=DEV)
    assert torch.equal(K.affine_relu_fwd(y, one, zero, True), y)


# ----------------------------------------------------------------------------- gamma / beta
GB_CASES = [
    # Co, Ci, k, stride, pad, B, H, W
    (512, 256, 3, 2, 1, 1, 8, 8),      # R: layer4.0.convbnrelu_1
    (512, 512, 3, 1, 1, 1, 4, 4),      # R: layer4.x 3x3
    (512, 256, 1, 2, 0, 1, 8, 8),      # R: layer4.0.shortcut (1x1, stride 2)
    (384, 192, 3, 1, 1, 1, 8, 8),      # A: features.4
    (64, 3, 5, 1, 2, 1, 32, 32),       # stem-like: K = 75 (not a multiple of 4) -> scalar path
    (16, 8, 3, 2, 1, 3, 9, 9),         # key batch 3 (mean over b)
    (24, 5, 3, 1, 0, 2, 7, 6),         # no padding, ragged
    (512, 256, 3, 2, 1, 1, 14, 14),    # ImageNet-shape layer4.0 (L = 49)
]


@pytest.mark.parametrize('cfg', GB_CASES)
def test_gamma_beta_fwd_bwd_dkey(K, cfg):
    co, ci, k, stride, pad, bk, h, w = cfg
    rs = np.random.RandomState(co + ci + k)
    wt = (rs.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / (co * k * k))).astype(np.float32)
    skey = rs.uniform(-1, 1, (bk, ci, h, w)).astype(np.float32)
    key = rs.uniform(-1, 1, (bk, ci, h, w)).astype(np.float32)
    m = K.pooled_patch_mean(dev(np.stack([skey, key])), k, k, stride, pad)
    s_s, n_s = npp.pooled_patch_sum(skey.astype(np.float64), k, k, stride, pad)
    s_b, _ = npp.pooled_patch_sum(key.astype(np.float64), k, k, stride, pad)
    close(host(m)[0], s_s / n_s, 'pooled mean (scale key)', 1e-12, 1e-13)
    close(host(m)[1], s_b / n_s, 'pooled mean (bias key)', 1e-12, 1e-13)

    gamma, beta = [host(t) for t in K.gamma_beta_fwd(dev(wt), m)]
    g64, b64 = npp.gamma_beta_fwd(wt.astype(np.float64), skey.astype(np.float64), key.astype(np.float64), stride, pad)
    # f64 accumulation + one rounding: within 1 ulp of the exact value (reference fp32 conv is ~1e-6 rel)
    assert np.all(np.abs(gamma - g64) <= 1.2e-7 * np.abs(g64) + 1e-12)
    assert np.all(np.abs(beta - b64) <= 1.2e-7 * np.abs(b64) + 1e-12)
    assert np.array_equal(np.sign(gamma), np.sign(g64.astype(np.float32))), 'signature bits'

    dg = rs.standard_normal(co).astype(np.float32)
    db = rs.standard_normal(co).astype(np.float32)
    dw = host(K.gamma_beta_bwd(dev(dg), dev(db), m, wt.shape))
    dw64, dsk64, dk64 = npp.gamma_beta_bwd(dg.astype(np.float64), db.astype(np.float64), wt.astype(np.float64),
                                           skey.astype(np.float64), key.astype(np.float64), stride, pad,
                                           need_dkey=True)
    close(dw, dw64, 'dW', 1e-5, 1e-6)
    dsk, dk = [host(t) for t in K.gamma_beta_dkey(dev(dg), dev(db), dev(wt), skey.shape, stride, pad)]
    close(dsk, dsk64, 'dskey', 1e-5, 1e-6)
    close(dk, dk64, 'dkey', 1e-5, 1e-6)


def test_gamma_beta_is_linear_in_the_key(K):
    """Property at full size (512x512x3x3): gamma(W, a*k1 + k2) == a*gamma(W,k1) + gamma(W,k2)."""
    rs = np.random.RandomState(3)
    wt = dev(rs.standard_normal((512, 512, 3, 3)) * 0.02)
    k1, k2 = rs.uniform(-1, 1, (1, 512, 4, 4)).astype(np.float32), rs.uniform(-1, 1, (1, 512, 4, 4)).astype(np.float32)
    mix = (np.float32(0.5) * k1 + k2).astype(np.float32)
    m = K.pooled_patch_mean(dev(np.stack([k1, k2, mix])), 3, 3, 1, 1)
    g1, g2 = K.gamma_beta_fwd(wt, m[:2].contiguous())
    g3, _ = K.gamma_beta_fwd(wt, torch.stack([m[2], m[2]]).contiguous())
    assert torch.allclose(g3, 0.5 * g1 + g2, rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------- sign loss
@pytest.mark.parametrize('c', [1, 5, 64, 256, 512, 1000])
def test_sign_loss_fwd_bwd(K, c):
    rs = np.random.RandomState(c)
    g = (rs.standard_normal(c) * 0.2).astype(np.float32)
    b = np.where(rs.uniform(size=c) < 0.5, -1.0, 1.0).astype(np.float32)
    if c >= 5:
        g[0] = 0.0                      # sign(0) = 0 never matches +-1 (trainer_private.py:50-53)
        g[1] = np.float32(0.1) * b[1]   # exactly on the hinge: relu'(0) = 0
        g[2] = -0.0
        g[3] = 1e-30
    for alpha in (0.1, 1.0):
        loss, acc, bits = K.sign_loss_fwd(dev(g), dev(b), alpha)
        l64, a64, bits64 = npp.sign_loss_fwd(g.astype(np.float64), b.astype(np.float64), alpha)
        z32 = (-b * g + np.float32(0.1))
        l_ref = (alpha * np.maximum(z32, 0).astype(np.float64)).sum() + 1e-5 * (g.astype(np.float64) ** 2).sum()
        assert abs(float(loss) - l_ref) <= 2e-6 * max(1.0, abs(l_ref))
        assert abs(float(loss) - float(l64)) <= 1e-5 * max(1.0, abs(float(l64)))
        assert float(acc) == pytest.approx(float(a64), abs=1e-7)
        assert np.array_equal(host(bits), bits64)
        dl = dev(np.array(1.7, dtype=np.float32))
        dg = host(K.sign_loss_bwd(dl, dev(g), dev(b), alpha))
        want = np.float32(1.7) * (np.where(z32 > 0, -alpha * b, 0) + 2e-5 * g)
        close(dg, want, 'sign_loss_bwd', 1e-6, 1e-7)


# ----------------------------------------------------------------------------- fused layer == unfused ops
@pytest.mark.parametrize('with_sign', [True, False])
@pytest.mark.parametrize('shape', [(128, 512, 4, 4, 512 * 9), (8, 64, 7, 7, 27), (4, 32, 16, 16, 288)])
def test_fused_layer_equals_unfused(K, shape, with_sign):
    n, c, h, w, kk = shape
    rs = np.random.RandomState(n + c)
    x, dy = dev(rs.standard_normal((n, c, h, w))), dev(rs.standard_normal((n, c, h, w)))
    wt = dev(rs.standard_normal((c, kk)) * 0.05)
    m = dev(rs.uniform(-1, 1, (2, kk)), torch.float64)
    b = dev(np.where(rs.uniform(size=c) < 0.5, -1.0, 1.0))
    dl = dev(np.array(0.5, dtype=np.float32))
    ex_g, ex_b = dev(rs.standard_normal(c) * 0.1), dev(rs.standard_normal(c) * 0.1)
    y, gamma, beta, loss, acc, bits = K.passport_fwd(x, wt, m, b if with_sign else None, ALPHA, True)
    g0, b0 = K.gamma_beta_fwd(wt, m)
    assert torch.equal(gamma, g0) and torch.equal(beta, b0)
    assert torch.equal(y, K.affine_relu_fwd(x, g0, b0, True))
    if with_sign:
        l0, a0, bits0 = K.sign_loss_fwd(g0, b, ALPHA)
        assert torch.equal(loss, l0) and torch.equal(acc, a0) and torch.equal(bits, bits0)
    dx, dw, dg, db = K.passport_bwd(dy, x, gamma, beta, m, b if with_sign else None, ALPHA,
                                    dl if with_sign else None, ex_g, ex_b, (c, kk), True)
    dx0, dg0, db0 = K.affine_relu_bwd(dy, x, g0, b0, True)
    dg0 = dg0 + ex_g
    if with_sign:
        dg0 = dg0 + K.sign_loss_bwd(dl, g0, b, ALPHA)
    db0 = db0 + ex_b
    assert torch.equal(dx, dx0)
    assert torch.allclose(dg, dg0, rtol=1e-6, atol=1e-6) and torch.allclose(db, db0, rtol=1e-6, atol=1e-6)
    assert torch.allclose(dw, K.gamma_beta_bwd(dg, db, m, (c, kk)), rtol=0, atol=0)


# ----------------------------------------------------------------------------- BatchNorm-fused layer
BN_SHAPES = [(128, 512, 4, 4, 4608), (64, 384, 8, 8, 1728), (6, 64, 32, 32, 27), (3, 16, 56, 56, 144),
             (5, 24, 7, 7, 75), (2, 3, 5, 3, 12), (9, 10, 2, 2, 40), (16, 64, 7, 7, 576), (4, 8, 9, 9, 72),
             # register-resident single pass: channel split over 4 / 2 workgroups with the in-launch exchange,
             # ragged last slice, forward-only fit (backward falls back), channel-owning, 2 and 8 channels per group
             (128, 64, 32, 32, 27), (128, 128, 16, 16, 576), (34, 64, 32, 32, 27), (130, 64, 32, 32, 27),
             (128, 256, 8, 8, 1152), (32, 512, 4, 4, 4608), (7, 16, 2, 2, 36)]


@pytest.fixture(params=['resident', 'two_pass'])
def bn_path(request):
    """Both implementations of the BatchNorm-fused layer: the register-resident single pass (taken whenever the
    shape fits) and the 3-launch form it falls back to."""
    from deepipr_amd import _lib
    _lib.set_resident(request.param == 'resident')
    yield request.param
    _lib.set_resident(True)


@pytest.mark.parametrize('mode', ['passport', 'passport_nosign', 'public', 'eval'])
@pytest.mark.parametrize('shape', BN_SHAPES)
def test_passport_bn_fused_fwd_bwd(K, shape, mode, bn_path):
    """deepipr_passport_bn_fwd/_bwd against a float64 ATen-style batch norm + the numpy passport oracle
    (tests/oracle_kernels.py) on the same inputs: y, running statistics, dx, dgamma, dbeta, dW."""
    from tests.oracle_kernels import OracleKernels
    O = OracleKernels()
    if bn_path == 'two_pass' and shape[0] * shape[1] * shape[2] * shape[3] > (1 << 22) and mode != 'passport':
        pytest.skip('large shapes: one mode is enough for the fallback path')
    n, c, h, w, kk = shape
    rs = np.random.RandomState(n * 7 + c)
    x = (rs.standard_normal((n, c, h, w)) * 1.7 + 0.3).astype(np.float32)
    dy = rs.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rs.standard_normal((c, kk)) * 0.05).astype(np.float32)
    m = rs.uniform(-1, 1, (2, kk))
    b = np.where(rs.uniform(size=c) < 0.5, -1.0, 1.0).astype(np.float32)
    g_in = (1 + 0.3 * rs.standard_normal(c)).astype(np.float32)
    b_in = (0.2 * rs.standard_normal(c)).astype(np.float32)
    rm0 = (0.1 * rs.standard_normal(c)).astype(np.float32)
    rv0 = (1 + 0.2 * rs.uniform(size=c)).astype(np.float32)
    dl = np.array(0.7, dtype=np.float32)
    training = mode != 'eval'
    public = mode == 'public'
    sign = mode in ('passport', 'eval')

    def run(kern, to):
        rm, rv = to(rm0.copy()), to(rv0.copy())
        nbt = to(np.array(3, dtype=np.int64), torch.int64)
        out = kern.passport_bn_fwd(to(x), None if public else to(wt), None if public else to(m, torch.float64),
                                   to(g_in) if public else None, to(b_in) if public else None,
                                   to(b) if sign else None, ALPHA, True, rm, rv, nbt, 0.1, 1e-5, training)
        y, table = out[0], out[1]
        back = kern.passport_bn_bwd(to(dy), to(x), table, None if public else to(m, torch.float64),
                                    to(b) if sign else None, ALPHA, to(dl) if sign else None, None, None,
                                    None if public else (c, kk), True, training)
        return out, back, rm, rv, nbt

    cpu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    (o_ref, b_ref, rm_r, rv_r, nbt_r) = run(O, cpu)
    (o_gpu, b_gpu, rm_g, rv_g, nbt_g) = run(K, dev)
    y_r, y_g = o_ref[0].numpy(), host(o_gpu[0])
    # a handful of elements may sit within rounding of the ReLU kink; everything else must agree to 2e-5
    bad = np.abs(y_g - y_r) > 2e-5 * (1 + np.abs(y_r))
    assert bad.mean() < 1e-5, bad.sum()
    close(host(o_gpu[1])[:, :4], o_ref[1].numpy()[:, :4], 'table', 2e-6, 2e-6)
    if training:
        close(host(rm_g), rm_r.numpy(), 'running_mean', 1e-6, 1e-6)
        close(host(rv_g), rv_r.numpy(), 'running_var', 1e-6, 1e-6)
        assert int(nbt_g) == int(nbt_r) == 4
    else:
        assert np.array_equal(host(rm_g), rm0) and np.array_equal(host(rv_g), rv0) and int(nbt_g) == 3
    if not public:
        close(host(o_gpu[2]), o_ref[2].numpy(), 'gamma', 2e-6, 2e-6)
        close(host(o_gpu[3]), o_ref[3].numpy(), 'beta', 2e-6, 2e-6)
    if sign:
        assert abs(float(o_gpu[4]) - float(o_ref[4])) < 2e-5 * max(1, abs(float(o_ref[4])))
        assert float(o_gpu[5]) == pytest.approx(float(o_ref[5]), abs=1e-7)
        assert np.array_equal(host(o_gpu[6]), o_ref[6].numpy())
    dx_r, dx_g = b_ref[0].numpy(), host(b_gpu[0])
    scale = np.abs(dx_r).max() + 1e-12
    bad = np.abs(dx_g - dx_r) > 1e-4 * scale
    assert bad.mean() < 1e-4, (bad.sum(), np.abs(dx_g - dx_r).max(), scale)
    for i, nm in ((2, 'dgamma'), (3, 'dbeta')):
        ref = b_ref[i].numpy()
        assert np.abs(host(b_gpu[i]) - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6), nm
    if not public:
        ref = b_ref[1].numpy()
        assert np.abs(host(b_gpu[1]) - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6)
    assert K.sync_timeouts() == 0


def test_resident_single_pass_is_deterministic_and_shares_its_exchange_words(K):
    """The in-launch exchange: layers with different slice counts interleaved on one stream (they share the
    exchange words, which every call advances by 64), repeated; results are bit-identical run to run and agree
    with the 3-launch form; no bounded wait ever expired."""
    from deepipr_amd import _lib
    shapes = [(128, 64, 32, 32), (128, 128, 16, 16), (64, 32, 32, 32), (128, 256, 8, 8), (33, 64, 16, 16)]
    rs = np.random.RandomState(3)
    data = []
    for n, c, h, w in shapes:
        data.append((dev(rs.standard_normal((n, c, h, w)) * 1.3 + 0.2), dev(rs.standard_normal((n, c, h, w))),
                     dev(1 + 0.3 * rs.standard_normal(c)), dev(0.2 * rs.standard_normal(c))))

    def sweep():
        outs = []
        for x, dy, g, b in data:
            c = x.shape[1]
            rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
            o = K.passport_bn_fwd(x, None, None, g, b, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
            bk = K.passport_bn_bwd(dy, x, o[1], None, None, 0.0, None, None, None, None, True, True)
            outs.append((o[0], o[1][:, :4].clone(), rm, rv, bk[0], bk[2], bk[3]))
        return outs

    first = sweep()
    for _ in range(3):
        again = sweep()
        for a, b in zip(first, again):
            for ta, tb in zip(a, b):
                assert torch.equal(ta, tb)
    assert K.sync_timeouts() == 0
    _lib.set_resident(False)
    try:
        ref = sweep()
    finally:
        _lib.set_resident(True)
    for a, b in zip(first, ref):
        for i, (ta, tb) in enumerate(zip(a, b)):
            scale = float(tb.abs().max()) + 1e-12
            diff = (ta - tb).abs()
            if i in (0, 4):                       # y / dx: isolated ReLU-kink flips allowed
                assert float((diff > 1e-4 * scale).float().mean()) < 1e-4
            else:
                assert float(diff.max()) <= 2e-5 * scale, i
    # in-situ timing sees one launch per direction on this path
    _lib.profile_enable(True)
    sweep()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    assert prof['bn_res_fwd'][1] == len(shapes) and prof['bn_res_bwd'][1] == len(shapes)
    assert prof['bn_stats'][1] == 0 and prof['bn_affine_bwd'][1] == 0


def test_fused_bn_layer_equals_unfused_layer_in_a_block(K):
    """PassportBlock with fuse_norm on/off: same outputs, gradients and running statistics."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    torch.manual_seed(4)
    np.random.seed(4)
    a = PassportBlock(64, 128, 3, 2, 1, kw).to(DEV)
    b = PassportBlock(64, 128, 3, 2, 1, kw).to(DEV)
    x = torch.randn(16, 64, 16, 16, device=DEV)
    with torch.no_grad():
        a(x)
    b.load_state_dict(a.state_dict())
    a.fuse_norm, b.fuse_norm = True, False
    outs = []
    for blk in (a, b):
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        (y.square().mean() + blk.sign_loss.loss).backward()
        outs.append((y.detach(), xi.grad, blk.weight.grad, blk.bn.running_mean.clone(), blk.bn.running_var.clone(),
                     blk.sign_loss.loss.detach()))
    for u, v, nm in zip(outs[0], outs[1], ('y', 'dx', 'dW', 'running_mean', 'running_var', 'sign_loss')):
        assert torch.allclose(u, v, rtol=2e-4, atol=2e-5 * float(v.abs().max() + 1e-9)), (nm, float((u - v).abs().max()))
    a.eval(), b.eval()
    with torch.no_grad():
        assert torch.allclose(a(x), b(x), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('shape', [(128, 64, 32, 32, 27), (128, 128, 16, 16, 576), (128, 512, 4, 4, 4608), (33, 256, 8, 8, 1152)])
@pytest.mark.parametrize('mode', ['passport', 'public'])
def test_fused_residual_tail_equals_separate_kernels(K, shape, mode):
    """out = relu(layer(x) + shortcut) folded into the single-pass norm kernels (forward: residual; backward:
    dy + dy2 masked by out, also returned as the shortcut's gradient) is bit-identical to the layer followed by
    deepipr_add_relu_fwd / deepipr_relu_bwd2."""
    n, c, h, w, kk = shape
    assert K.bn_resident(n, c, h * w) == 3
    rs = np.random.RandomState(c + n)
    x, r, g1, g2 = [dev(rs.standard_normal((n, c, h, w))) for _ in range(4)]
    wt = dev(rs.standard_normal((c, kk)) * 0.05)
    m = dev(rs.uniform(-1, 1, (2, kk)), torch.float64)
    b = dev(np.where(rs.uniform(size=c) < 0.5, -1.0, 1.0))
    gi, bi = dev(1 + 0.3 * rs.standard_normal(c)), dev(0.2 * rs.standard_normal(c))
    dl = dev(np.array(0.7))
    public = mode == 'public'

    def fwd(residual):
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        return K.passport_bn_fwd(x, None if public else wt, None if public else m, gi if public else None,
                                 bi if public else None, None if public else b, ALPHA, True, rm, rv, None, 0.1, 1e-5,
                                 True, residual=residual)

    def bwd(table, dy, **kw):
        return K.passport_bn_bwd(dy, x, table, None if public else m, None if public else b, ALPHA,
                                 None if public else dl, None, None, None if public else (c, kk), True, True, **kw)

    plain, fused = fwd(None), fwd(r)
    out = K.add_relu_fwd(plain[0], r)
    assert torch.equal(fused[0], out)
    assert torch.equal(fused[1], plain[1])
    for second in (g2, None):
        d = K.relu_bwd(g1, out, second)
        ref = bwd(plain[1], d)
        got = bwd(fused[1], g1, dy2=second, tail_out=fused[0])
        assert torch.equal(got[4], d)
        for a_, b_ in zip(got[:4], ref):
            if a_ is not None:
                assert torch.equal(a_, b_)


def test_residual_blocks_use_the_fused_tail():
    """A ResNet18 train step on the GPU: the 8 block tails go through the norm kernels, no k_add_relu / k_relu_bwd
    launches are left (DEEPIPR_TAIL_FUSION=0 would bring them back); the two plain projection blocks (layer2.0, layer3.0)
    run their last two norm layers + tail as ONE launch per direction (DEEPIPR_NO_DUAL_TAIL=1: two), so the 20 fused
    layer calls of a step are 18 launches each way."""
    import os
    if os.environ.get('DEEPIPR_TAIL_FUSION', '1') == '0':
        pytest.skip('tail fusion switched off')
    from deepipr_amd import _lib
    from deepipr_amd.experiments.trainer import train_step_v1
    prod, _ref, x, y = _fullsize_pair(False, 128, 10)     # config R: every norm layer above the ConvBlock fusion size
    opt = torch.optim.SGD(prod.parameters(), lr=0.0)
    train_step_v1(prod, opt, x.to(DEV), y.to(DEV))
    _lib.profile_enable(True)
    train_step_v1(prod, opt, x.to(DEV), y.to(DEV))
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    assert prof['add_relu'][1] == 0, prof['add_relu']
    want = 20 if os.environ.get('DEEPIPR_NO_DUAL_TAIL') == '1' else 18
    assert prof['bn_res_fwd'][1] == want and prof['bn_res_bwd'][1] == want, (prof['bn_res_fwd'], prof['bn_res_bwd'])


# ----------------------------------------------------------------------------- GroupNorm / InstanceNorm-fused layer
GN_SHAPES = [  # n, c, h, w, groups, kk
    (64, 384, 8, 8, 24, 1728),      # AlexNet GN (o // 16 groups): 16 channels x 64 = 256 units, one workgroup per chunk
    (64, 256, 8, 8, 256, 2304),     # AlexNet IN: 16 units, 16 chunks per wavefront
    (128, 512, 4, 4, 32, 4608),     # ResNet layer4 GN: 64 units
    (128, 512, 4, 4, 512, 4608),    # ResNet layer4 IN: 4 units (4 lanes per chunk)
    (6, 64, 32, 32, 4, 27),         # stem GN: 4096 units -> 1024 lanes x 4
    (5, 64, 32, 32, 64, 27),        # stem IN: 256 units
    (3, 48, 12, 12, 3, 75),         # 576 units (not a power of two): 1024 lanes, 448 idle
    (2, 6, 2, 2, 6, 12), (7, 12, 6, 2, 2, 40),   # tiny / ragged last workgroup
]


@pytest.mark.parametrize('mode', ['passport', 'passport_nosign', 'public', 'plain'])
@pytest.mark.parametrize('shape', GN_SHAPES)
def test_passport_gn_fused_fwd_bwd(K, shape, mode):
    """deepipr_passport_gn_fwd/_bwd against float64 group statistics + the numpy passport oracle
    (tests/oracle_kernels.py): y, stats, dx, dgamma, dbeta, dW.  'plain' = no gamma/beta at all (InstanceNorm2d
    without affine in a ConvBlock)."""
    from tests.oracle_kernels import OracleKernels
    O = OracleKernels()
    n, c, h, w, groups, kk = shape
    assert K.gn_supported(n, c, h * w, groups) and O.gn_supported(n, c, h * w, groups)
    rs = np.random.RandomState(n * 7 + c + groups)
    x = (rs.standard_normal((n, c, h, w)) * 1.7 + 0.3).astype(np.float32)
    dy = rs.standard_normal((n, c, h, w)).astype(np.float32)
    wt = (rs.standard_normal((c, kk)) * 0.05).astype(np.float32)
    m = rs.uniform(-1, 1, (2, kk))
    b = np.where(rs.uniform(size=c) < 0.5, -1.0, 1.0).astype(np.float32)
    g_in = (1 + 0.3 * rs.standard_normal(c)).astype(np.float32)
    b_in = (0.2 * rs.standard_normal(c)).astype(np.float32)
    dl = np.array(0.7, dtype=np.float32)
    wless = mode in ('public', 'plain')
    sign = mode == 'passport'

    def run(kern, to):
        gi = to(g_in) if mode == 'public' else None
        bi = to(b_in) if mode == 'public' else None
        out = kern.passport_gn_fwd(to(x), None if wless else to(wt), None if wless else to(m, torch.float64), gi, bi,
                                   to(b) if sign else None, ALPHA, True, groups, 1e-5)
        used_g, used_b = (out[2], out[3]) if not wless else (gi, bi)
        back = kern.passport_gn_bwd(to(dy), to(x), out[1], used_g, used_b, None if wless else to(m, torch.float64),
                                    to(b) if sign else None, ALPHA, to(dl) if sign else None, None, None,
                                    None if wless else (c, kk), True, groups)
        return out, back

    cpu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    o_ref, b_ref = run(O, cpu)
    o_gpu, b_gpu = run(K, dev)
    y_r, y_g = o_ref[0].numpy(), host(o_gpu[0])
    bad = np.abs(y_g - y_r) > 2e-5 * (1 + np.abs(y_r))
    assert bad.mean() < 1e-5, bad.sum()
    close(host(o_gpu[1]), o_ref[1].numpy(), 'stats', 5e-6, 5e-6)
    if not wless:
        close(host(o_gpu[2]), o_ref[2].numpy(), 'gamma', 2e-6, 2e-6)
        close(host(o_gpu[3]), o_ref[3].numpy(), 'beta', 2e-6, 2e-6)
    if sign:
        assert abs(float(o_gpu[4]) - float(o_ref[4])) < 2e-5 * max(1, abs(float(o_ref[4])))
        assert float(o_gpu[5]) == pytest.approx(float(o_ref[5]), abs=1e-7)
        assert np.array_equal(host(o_gpu[6]), o_ref[6].numpy())
    dx_r, dx_g = b_ref[0].numpy(), host(b_gpu[0])
    scale = np.abs(dx_r).max() + 1e-12
    bad = np.abs(dx_g - dx_r) > 1e-4 * scale
    assert bad.mean() < 1e-4, (bad.sum(), np.abs(dx_g - dx_r).max(), scale)
    for i, nm in ((2, 'dgamma'), (3, 'dbeta')):
        ref = b_ref[i].numpy()
        assert np.abs(host(b_gpu[i]) - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6), nm
    if not wless:
        ref = b_ref[1].numpy()
        assert np.abs(host(b_gpu[1]) - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6)
    # deterministic
    o2, b2 = run(K, dev)
    assert torch.equal(o2[0], o_gpu[0]) and torch.equal(b2[0], b_gpu[0]) and torch.equal(b2[2], b_gpu[2])


def test_gn_unsupported_shapes_are_refused_without_side_effects(K):
    """Chunks beyond the register budget and planes that are not a multiple of 4 floats: gn_supported() says no,
    the entry point returns DEEPIPR_EUNSUPPORTED, and the layer falls back to the library norm + affine kernels."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    assert not K.gn_supported(8, 64, 56 * 56, 4) and not K.gn_supported(8, 64, 49, 4) and not K.gn_supported(8, 60, 16, 7)
    x = torch.randn(2, 64, 56, 56, device=DEV)
    with pytest.raises(RuntimeError, match='does not fit the fused form'):
        K.passport_gn_fwd(x, None, None, None, None, None, 0.0, True, 4, 1e-5)
    torch.manual_seed(0)
    blk = PassportBlock(8, 16, 3, 1, 1, {'norm_type': 'gn', 'key_type': 'random', 'sign_loss': 0.1}).to(DEV)
    xin = torch.randn(2, 8, 7, 7, device=DEV, requires_grad=True)            # 7x7 planes: unfused path
    y = blk(xin)
    (y.sum() + blk.sign_loss.loss).backward()
    assert torch.isfinite(xin.grad).all() and blk.weight.grad is not None


@pytest.mark.parametrize('norm', ['gn', 'in'])
def test_gn_and_in_layers_take_the_fused_kernels(norm):
    """AlexNet with norm_type gn / in: all five conv layers (2 ConvBlocks, 3 passport layers) go through
    k_gn_fwd / k_gn_bwd -- counted by the in-situ profile, so a silent fallback to the library norm would show."""
    from deepipr_amd import _lib
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.alexnet_passport import AlexNetPassport
    from oracle.cases import alexnet_config
    kw = construct_passport_kwargs_from_dict({'passport_config': alexnet_config(), 'norm_type': norm,
                                              'key_type': 'random', 'sl_ratio': ALPHA})
    torch.manual_seed(0)
    np.random.seed(0)
    net = AlexNetPassport(3, 10, kw).to(DEV).train()
    x = torch.randn(8, 3, 32, 32, device=DEV)
    net(x)                                                    # materialise keys
    _lib.profile_enable(True)
    out = net(x)
    sl = sum(m.sign_loss.loss for m in net.modules() if hasattr(m, 'sign_loss') and m.sign_loss is not None)
    (out.sum() + sl).backward()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    assert prof['gn_fwd'][1] == 5 and prof['gn_bwd'][1] == 5
    assert prof['affine_fwd'][1] == 0 and prof['affine_bwd'][1] == 0
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


# ----------------------------------------------------------------------------- golden fixtures (real reference)
@pytest.mark.parametrize('fuse_norm', [True, False])
@pytest.mark.parametrize('name', list(CASES))
def test_model_cases_match_reference_goldens(K, name, fuse_norm, golden_dir):
    if not fuse_norm and CASES[name]['norm'] == 'none':
        pytest.skip('no norm, nothing to fuse')
    """Whole nets + one optimisation step through the product trainers on the GPU, against outputs of the
    real reference.  Logits / losses / gamma / beta within 1e-4, signature bits identical."""
    gold = load_golden(golden_dir, name)
    got = runner.collect(name, ProductImpl(DEV, fuse_norm=fuse_norm))
    loose = ('grad/', 'post/', 'stat/', 'logits_eval/', 'train/acc')      # checked below with their own bars
    compare_case(got, gold, rtol=1e-4, atol=1e-4, skip_prefixes=loose + ('ctor_b/',))
    # Bars per key family = 10 x the worst deviation measured over all 17 cases, fused norm on and off
    # (tools/golden_margins.py -> profiles/r05_golden_margins.json with the Winograd kernels: grad 1.1e-3 of scale,
    # logits_eval 4.8e-5, post 4.1e-6, stat 8e-8; round 4: 7.5e-4 / 3.2e-5 / 4.1e-6 / 1.1e-7).  Round 3 held all four at rtol 5e-3 / atol 1e-3.  The gradient digests
    # keep that bar: golden batches are 2-8 images, so every norm layer's backward divides by a batch variance of a
    # handful of samples and one flipped ReLU mask moves a whole digest; the tight whole-net gradient bars (1e-4 of
    # scale, kinks gated) are test_whole_net_backward_within_1e4_with_relu_kinks_gated's, at the real batch sizes.
    bars = {'grad/': (5e-3, 1e-3), 'logits_eval/': (1e-4, 3e-4), 'post/': (1e-4, 5e-5), 'stat/': (1e-4, 1e-5)}
    for k in gold:
        for prefix, (rtol, atol) in bars.items():
            if k.startswith(prefix):
                close(got[k], gold[k], k, rtol=rtol, atol=atol)
        if k.startswith('ctor_b/'):
            assert np.array_equal(got[k], gold[k]), k


@pytest.mark.miopen_pinned
@pytest.mark.parametrize('private', [False, True])
def test_tail_fusion_is_bit_identical_at_model_level(private, monkeypatch):
    """The residual-tail fusion (default; DEEPIPR_TAIL_FUSION=0 switches it off) against the separate tail kernels on whole nets, MIOpen
    pinned to its deterministic algorithms: logits and every parameter gradient bit-identical.

    No retry: a mismatch fails.  (Round 2 saw ONE mismatch in seven runs of the whole GPU suite and never reproduced it
    -- 0 in 2 440 fresh-process steps; round 3: 0 in every session-end run of tests/test_zz_session_end_gpu.py, which
    repeats this pair 2 x 40 times in the state a full session leaves behind and LOCALISES a mismatch instead of retrying
    (first differing module in each direction, same-form repeat, kernel lists), and 0 differing results in 17 600
    repeated vendor-convolution calls of the nets' shapes in pinned mode, tools/conv_determinism.py;
    profiles/r03_determinism.md.)"""
    bench, det = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    try:
        n, ncls = (64, 100) if private else (128, 10)
        prod, _ref, x, y = _fullsize_pair(private, n, ncls)
        x, y = x.to(DEV), y.to(DEV)
        ce = torch.nn.functional.cross_entropy
        state = {k: v.clone() for k, v in prod.state_dict().items()}

        def step(flag):
            monkeypatch.setenv('DEEPIPR_TAIL_FUSION', flag)
            prod.zero_grad(set_to_none=True)
            if private:
                outs = [prod(x, ind=0), prod(x, ind=1)]
                loss = ce(outs[0], y) + ce(outs[1], y)
                loss = loss + sum(m.sign_loss_private.loss for m in prod.modules() if hasattr(m, 'sign_loss_private'))
            else:
                outs = [prod(x)]
                loss = ce(outs[0], y) + sum(m.sign_loss.loss for m in prod.modules()
                                            if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
            loss.backward()
            got = {'logits%d' % i: o.detach().clone() for i, o in enumerate(outs)}
            got.update({k: p.grad.clone() for k, p in prod.named_parameters() if p.grad is not None})
            prod.load_state_dict(state)                      # norm running statistics back to where they were
            return got

        def differing(a, b):
            assert set(a) == set(b)
            return {k: float((a[k] - b[k]).abs().max()) for k in a if not torch.equal(a[k], b[k])}

        separate, fused = step('0'), step('1')
        diff = differing(separate, fused)
        assert not diff, ('%d of %d tensors differ between the fused and the separate tail (largest: %s)'
                          % (len(diff), len(fused), sorted(diff.items(), key=lambda kv: -kv[1])[:4]))
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bench, det


# ----------------------------------------------------------------------------- full size vs the oracle
def _fullsize_pair(private, n, ncls):
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.resnet_passport import ResNet18Passport
    from deepipr_amd.models.resnet_passport_private import ResNet18Private
    cfg = resnet18_config()
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random',
                                              'sl_ratio': ALPHA})
    torch.manual_seed(0)
    np.random.seed(0)
    prod = (ResNet18Private if private else ResNet18Passport)(num_classes=ncls, passport_kwargs=kw).to(DEV)
    ref = torch_ref.resnet18_ref(num_classes=ncls, passport_kwargs=torch_ref.passport_kwargs_from_config(
        cfg, 'bn', 'random', ALPHA), private=private)
    x, y = patterns.batch(n, 3, 32, 32, ncls)
    prod.train()
    ref.train()
    with torch.no_grad():
        prod(x.to(DEV))
        ref(x)
    patterns.fill_state(prod)
    patterns.fill_state(ref)
    return prod, ref, x, y


@pytest.mark.miopen_pinned
@pytest.mark.parametrize('fuse_norm', [True, False])
@pytest.mark.parametrize('private', [False, True])
def test_product_equals_stock_aten_on_gpu(private, fuse_norm):
    """Same GPU, same MIOpen convs: the product (HIP passport kernels, with and without the fused BatchNorm)
    against the oracle's stock-ATen composition moved to the GPU.  MIOpen is pinned to its deterministic
    default algorithms (find mode picks different, differently-rounded algorithms per model instance, which
    measured as up to 2e-2 relative noise on small early-layer gradients); with that removed the hand-written
    kernels are isolated: logits 2e-5, every parameter gradient within 2e-4 of its scale (measured 3.5e-5)."""
    bench, det = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
    import os
    if os.environ.get('DEEPIPR_TEST_KEEP_FIND_MODE') != '1':       # triage switch, see DESIGN.md 7 (tail fusion)
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    try:
        n, ncls = (64, 100) if private else (128, 10)
        prod, ref, x, y = _fullsize_pair(private, n, ncls)
        for m in prod.modules():
            if hasattr(m, 'fuse_norm'):
                m.fuse_norm = fuse_norm
        ref = ref.to(DEV)
        x, y = x.to(DEV), y.to(DEV)
        ce = torch.nn.functional.cross_entropy
        if private:
            p0, p1, r0, r1 = prod(x, ind=0), prod(x, ind=1), ref(x, ind=0), ref(x, ind=1)
            assert torch.allclose(p0, r0, rtol=2e-5, atol=2e-5) and torch.allclose(p1, r1, rtol=2e-5, atol=2e-5)
            lp, lr = ce(p0, y) + ce(p1, y), ce(r0, y) + ce(r1, y)
            sp = sum(m.sign_loss_private.loss for m in prod.modules() if hasattr(m, 'sign_loss_private'))
        else:
            out_p, out_r = prod(x), ref(x)
            assert torch.allclose(out_p, out_r, rtol=2e-5, atol=2e-5), (out_p - out_r).abs().max()
            lp, lr = ce(out_p, y), ce(out_r, y)
            sp = sum(m.sign_loss.loss for m in prod.modules() if getattr(m, 'sign_loss', None) is not None
                     and hasattr(m, 'conv'))
        sr = sum(m.loss for m in torch_ref.sign_losses(ref))
        assert abs(float(sp.detach()) - float(sr.detach())) < 2e-5 * max(1.0, abs(float(sr.detach())))
        (lp + sp).backward()
        (lr + sr).backward()
        # Gradients.  An activation that sits within fp32 rounding of a ReLU kink may be masked differently by
        # two correct implementations (tests/triage/debug_hooks.py found exactly ONE such element of 2.6 M in the
        # private case: one flipped mask => 21 % of max|dx| at that element, 2e-2 on the 4608 weights of its
        # output channel, ~1e-3 on everything upstream).  So: 99 % of every gradient's elements within 5e-3
        # of its scale, and no element further than what a couple of flips can explain.
        gp = dict(prod.named_parameters())
        for name, p in ref.named_parameters():
            a, b = gp[name].grad, p.grad
            scale = float(b.abs().max()) + 1e-12
            diff = (a - b).abs()
            # (one flip in layer4.1 also perturbs every gradient upstream of it, layer4.0 included, at the
            # 1e-3 level -- so the bound is global, not per layer; the op-level tests above are the tight ones)
            n_bad = int((diff > 5e-3 * scale + 1e-7).sum())
            frac_bad = n_bad / diff.numel()
            # (per-channel vectors have 64-512 elements: a few channels downstream of a flipped mask may exceed the
            # 5e-3 bar, more so when MIOpen's immediate mode picks Winograd kernels; the max-error bound below and
            # the op-level tests are the tight ones)
            assert frac_bad <= 0.01 or n_bad <= 3, (name, frac_bad, n_bad)
            assert float(diff.max()) <= 0.25 * scale + 1e-7, (name, float(diff.max()), scale)
        for (na, ba), (nb, bb) in zip(prod.named_buffers(), ref.named_buffers()):
            if na.endswith(('running_mean', 'running_var')):
                assert torch.allclose(ba, bb, rtol=1e-5, atol=1e-6), na
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bench, det


def test_resnet18_v1_config_R_full_size_step():
    """BASELINE config 1: ResNet18 V1, CIFAR10 shapes, batch 128 -- one train step on the GPU vs the CPU
    oracle: logits and sign loss within 1e-4, signature bits exact, post-step weights close."""
    torch.set_num_threads(max(1, torch.get_num_threads()))
    from deepipr_amd.experiments.trainer import train_step_v1
    prod, ref, x, y = _fullsize_pair(False, 128, 10)
    logits = []
    h = prod.register_forward_hook(lambda m, i, o: logits.append(o.detach().cpu()))
    opt_p = torch.optim.SGD(prod.parameters(), **SGD)
    opt_r = torch.optim.SGD(ref.parameters(), **SGD)
    loss, sign_loss, _ = train_step_v1(prod, opt_p, x.to(DEV), y.to(DEV))
    h.remove()
    out = torch_ref.v1_step(ref, opt_r, x, y)
    assert torch.allclose(logits[0], out['pred'], rtol=1e-4, atol=1e-4), (logits[0] - out['pred']).abs().max()
    assert abs(float(loss) - float(out['loss'])) < 1e-4
    assert abs(float(sign_loss) - float(out['sign_loss'])) < 1e-4
    rep_p = {n: m for n, m in prod.named_modules() if hasattr(m, 'sign_loss') and m.sign_loss is not None}
    for name, m in ref.named_modules():
        if isinstance(m, torch_ref.PassportLayerRef):
            g_ref = m.sign_loss.scale_cache.detach().view(-1)
            g_gpu = rep_p[name].sign_loss.scale_cache.detach().view(-1).cpu()
            assert torch.allclose(g_gpu, g_ref, rtol=1e-4, atol=1e-6)
            assert torch.equal(g_gpu.sign(), g_ref.sign()), name          # signature bits, pre-step
    sd_p, sd_r = prod.state_dict(), ref.state_dict()
    for k in sd_r:
        if sd_r[k].dtype.is_floating_point:
            assert torch.allclose(sd_p[k].cpu(), sd_r[k], rtol=1e-3, atol=2e-4), k


def test_resnet18_v2_private_full_size_step():
    """Config P per-GPU shard (batch 32, CIFAR100 shapes): dual forward / one backward."""
    from deepipr_amd.experiments.trainer_private import DualBranch, TesterPrivate, train_step_v23
    prod, ref, x, y = _fullsize_pair(True, 32, 100)
    dual = DualBranch(prod)
    opt_p = torch.optim.SGD(prod.parameters(), **SGD)
    opt_r = torch.optim.SGD(ref.parameters(), **SGD)
    logits = []
    h = prod.register_forward_hook(lambda m, i, o: logits.append(o.detach().cpu()))
    loss, sign_loss, _, _ = train_step_v23(dual, opt_p, x.to(DEV), y.to(DEV))
    h.remove()
    out = torch_ref.v23_step(ref, opt_r, x, y)
    assert torch.allclose(logits[0], out['pred_public'], rtol=1e-4, atol=1e-4)
    assert torch.allclose(logits[1], out['pred_private'], rtol=1e-4, atol=1e-4)
    assert abs(float(loss) - float(out['loss'])) < 1e-4 and abs(float(sign_loss) - float(out['sign_loss'])) < 1e-4
    sig_p = TesterPrivate(prod, torch.device(DEV), verbose=False).test_signature()
    sig_r = torch_ref.signature_report(ref)
    assert set(sig_p) == set(sig_r)
    for k in sig_r:
        assert sig_p[k] == pytest.approx(sig_r[k][1], abs=1e-7), k         # detection rates identical


def test_signature_embeds_and_reads_back_bit_exact():
    """Train a single passport layer's gamma towards an ASCII signature with the sign loss only, then
    read it back through sign(gamma): the decoded text must be exact (README.md:92-94 of the reference)."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    torch.manual_seed(1)
    np.random.seed(1)
    text = 'MI355X!!'
    blk = PassportBlock(16, 64, 3, 1, 1, {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 1.0, 'b': text}).to(DEV)
    x = torch.randn(8, 16, 8, 8, device=DEV)
    opt = torch.optim.SGD(blk.parameters(), lr=0.05, momentum=0.9)
    for _ in range(200):
        opt.zero_grad()
        blk(x)
        blk.sign_loss.loss.backward()
        opt.step()
    with torch.no_grad():
        bits = blk.get_scale().view(-1).sign().cpu().numpy()
    assert npp.decode_signature(bits) == text
    assert float(blk.sign_loss.acc) == 1.0


@pytest.mark.miopen_pinned
def test_graphed_step_equals_eager_step():
    """hipGraph-captured V2 step (dual forward, fused-BN passport kernels inside the graph) replays to the same
    weights as the eager step."""
    from deepipr_amd.experiments.graph_step import GraphedTrainStep
    from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23
    results = []
    for graphed in (False, True):
        prod, _ref, x, y = _fullsize_pair(True, 32, 100)
        x, y = x.to(DEV), y.to(DEV)
        dual = DualBranch(prod)
        opt = torch.optim.SGD(prod.parameters(), **SGD)
        bench, det = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
        try:
            if graphed:
                snap = {k: v.clone() for k, v in prod.state_dict().items()}
                g = GraphedTrainStep(train_step_v23, dual, opt, x, y, warmup=2)
                prod.load_state_dict(snap)                    # undo the warm-up / capture steps
                for st in opt.state.values():
                    st['momentum_buffer'].zero_()
                for _ in range(3):
                    out = g(x, y)
            else:
                for _ in range(3):
                    out = train_step_v23(dual, opt, x, y)
            torch.cuda.synchronize()
        finally:
            torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bench, det
        results.append(({k: v.clone() for k, v in prod.state_dict().items()}, [float(o) for o in out]))
    (sd_e, out_e), (sd_g, out_g) = results
    for a, b in zip(out_e, out_g):
        assert a == pytest.approx(b, rel=1e-4, abs=1e-5)
    states_close(sd_e, sd_g, what='eager vs graphed')


@pytest.mark.miopen_pinned
def test_trainer_graph_mode_equals_eager_epoch():
    """Trainer(graph=True): captured on the first batch of each shape (without advancing training) and replayed -- the ragged
    batch gets a graph of its own (StepRunner._others), captured on its first occurrence and replayed on its second -- same
    epoch result and weights as the eager Trainer."""
    from deepipr_amd.experiments.trainer import Trainer
    from deepipr_amd.flat_sgd import FlatSGD
    outs = []
    for graph, flat in ((False, False), (True, False), (True, True)):
        prod, _ref, x, y = _fullsize_pair(False, 32, 10)
        x, y = x.to(DEV), y.to(DEV)
        loader = [(x, y), (x.flip(0), y.flip(0)), (x[:20], y[:20]), (x * 0.5, y), (x[:20].flip(0), y[:20].flip(0))]
        opt = (FlatSGD if flat else torch.optim.SGD)(prod.parameters(), **SGD)
        bench, det = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
        try:
            res = Trainer(prod, opt, None, torch.device(DEV), graph=graph).train(1, loader)
        finally:
            torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bench, det
        outs.append((res, {k: v.clone() for k, v in prod.state_dict().items()}))
    (re, se) = outs[0]
    for (rg, sg) in outs[1:]:
        for k in ('loss', 'sign_loss', 'sign_acc', 'acc'):
            assert re[k] == pytest.approx(rg[k], rel=1e-4, abs=1e-5), k
        states_close(se, sg, what='eager epoch vs graph-mode epoch')
        for k in se:
            if not se[k].dtype.is_floating_point:
                assert torch.equal(se[k], sg[k]), k


def test_entry_points_run_on_the_gpu(tmp_path, monkeypatch):
    """train_v1.py (shuffle keys drawn from the trigger set; hipGraph replay is the default on the GPU) and
    train_v23.py --train-backdoor end to end on synthetic data."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.chdir(root)
    sys.path.insert(0, root)
    import train_v1
    import train_v23
    out = train_v1.main(['--arch', 'resnet', '--train-passport', '--key-type', 'shuffle', '--epochs', '2',
                         '--passport-config', 'passport_configs/resnet18_passport.json', '--batch-size', '64',
                         '--synthetic-samples', '256', '--use-trigger-as-passport', '--logdir', str(tmp_path)])
    h = out['history']
    assert len(h) == 2 and all(np.isfinite(r['train_loss']) for r in h) and h[0]['train_sign_loss'] > 0
    assert h[1]['train_sign_loss'] < h[0]['train_sign_loss']          # the hinge is being driven down
    out = train_v23.main(['--arch', 'alexnet', '--train-backdoor', '--key-type', 'random', '--epochs', '1',
                          '--batch-size', '32', '--synthetic-samples', '128', '--dataset', 'cifar100',
                          '--logdir', str(tmp_path)])
    assert 'valid_s_private_features.4' in out['history'][0]


def test_reference_checkpoints_on_gpu(golden_dir):
    """Reference-written state_dicts -> strict load -> .to(gpu) -> the reference's eval outputs (HIP kernels)."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
    gold = load_golden(golden_dir, 'blocks')
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    x = dev(gold['ckpt_in/x'])
    sd = {k[len('ckpt_v1/'):]: torch.from_numpy(np.array(v)) for k, v in gold.items() if k.startswith('ckpt_v1/')}
    for fuse in (True, False):
        blk = PassportBlock(4, 16, 3, 1, 1, kw)
        blk.fuse_norm = fuse
        blk.load_state_dict(sd, strict=True)
        blk = blk.to(DEV).eval()
        with torch.no_grad():
            close(host(blk(x)), gold['ckpt_v1_out/y'], 'v1 eval', 1e-4, 1e-5)
    sd = {k[len('ckpt_private/'):]: torch.from_numpy(np.array(v)) for k, v in gold.items()
          if k.startswith('ckpt_private/')}
    blk = PassportPrivateBlock(4, 16, 3, 1, 1, kw)
    blk.load_state_dict(sd, strict=True)
    blk = blk.to(DEV).eval()
    with torch.no_grad():
        close(host(blk(x, ind=0)), gold['ckpt_private_out/y0'], 'public eval', 1e-4, 1e-5)
        close(host(blk(x, ind=1)), gold['ckpt_private_out/y1'], 'private eval', 1e-4, 1e-5)


@pytest.mark.miopen_pinned
def test_flat_sgd_equals_torch_sgd_on_gpu(K):
    """FlatSGD (flat buffers + the fused HIP SGD kernel) against torch.optim.SGD over four train steps, and the
    kernel alone against the update rule in float64."""
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.flat_sgd import FlatSGD
    rs = np.random.RandomState(0)
    n = 1000003                                              # odd length: scalar tail path
    p0, g0, b0 = [rs.standard_normal(n).astype(np.float32) for _ in range(3)]
    for nn in (n, 1 << 20):
        p, g, b = dev(p0[:nn]), dev(g0[:nn]), dev(b0[:nn])
        K.sgd_momentum_step(p, g, b, 0.05, 0.9, 1e-4, 0.5)
        d = 0.5 * g0[:nn].astype(np.float64) + 1e-4 * p0[:nn]
        bw = 0.9 * b0[:nn] + d
        close(host(b), bw, 'momentum buffer', 1e-6, 1e-6)
        close(host(p), p0[:nn] - 0.05 * bw, 'parameter', 1e-6, 1e-6)
    finals = []
    bench, det = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    try:
        for flat in (False, True):
            prod, _ref, x, y = _fullsize_pair(False, 32, 10)
            x, y = x.to(DEV), y.to(DEV)
            opt = (FlatSGD if flat else torch.optim.SGD)(prod.parameters(), **SGD)
            for i in range(4):
                train_step_v1(prod, opt, x if i % 2 == 0 else x.flip(0), y if i % 2 == 0 else y.flip(0))
            finals.append({k: v.clone() for k, v in prod.state_dict().items()})
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bench, det
    states_close(finals[0], finals[1], what='eager vs replayed')


def test_convblock_fused_norm_equals_library_ops():
    """ConvBlock on the GPU: fused norm+affine+ReLU kernels (default above 1M elements) against MIOpen batch norm +
    ATen ReLU on the same block: output, input/weight/bn gradients, running statistics; train and eval."""
    from deepipr_amd.models.layers.conv2d import ConvBlock
    torch.manual_seed(2)
    a, b = ConvBlock(32, 64, 3, 1, 1).to(DEV), ConvBlock(32, 64, 3, 1, 1).to(DEV)
    b.load_state_dict(a.state_dict())
    with torch.no_grad():
        for blk in (a, b):
            blk.bn.weight.copy_(1 + 0.3 * torch.randn(64, device=DEV, generator=torch.Generator(DEV).manual_seed(1)))
            blk.bn.bias.copy_(0.2 * torch.randn(64, device=DEV, generator=torch.Generator(DEV).manual_seed(2)))
    a.fuse_norm, b.fuse_norm = True, False
    x = torch.randn(64, 32, 32, 32, device=DEV)                     # 4.2M output elements: fused path taken
    cot = torch.randn(64, 64, 32, 32, device=DEV)
    res = []
    for blk in (a, b):
        xi = x.clone().requires_grad_(True)
        y = blk(xi)
        (y * cot).sum().backward()
        res.append((y.detach(), xi.grad, blk.conv.weight.grad, blk.bn.weight.grad, blk.bn.bias.grad,
                    blk.bn.running_mean.clone(), blk.bn.running_var.clone(), blk.bn.num_batches_tracked.clone()))
    names = ('y', 'dx', 'dW', 'd bn.weight', 'd bn.bias', 'running_mean', 'running_var', 'num_batches_tracked')
    for u, v, nm in zip(res[0], res[1], names):
        u, v = u.float(), v.float()
        assert float((u - v).abs().max()) <= 2e-4 * float(v.abs().max()) + 1e-6, nm
    a.eval(), b.eval()
    with torch.no_grad():
        assert torch.allclose(a(x), b(x), rtol=1e-4, atol=1e-5)
    small = torch.randn(2, 32, 8, 8, device=DEV)                    # below FUSE_MIN_ELEMENTS: library path either way
    a.train(), b.train()
    assert torch.allclose(a(small), b(small), rtol=1e-4, atol=1e-5)


def test_in_situ_profile_counts_launches(K):
    """deepipr_profile_enable / _read: per-dispatch events, counted per kernel, resumable."""
    from deepipr_amd import _lib
    x = torch.randn(8, 16, 8, 8, device=DEV)
    g, b = torch.randn(16, device=DEV), torch.randn(16, device=DEV)
    _lib.profile_enable(1)
    for _ in range(3):
        K.affine_relu_fwd(x, g, b, True)
    _lib.profile_enable(0)
    K.affine_relu_fwd(x, g, b, True)                                # not counted: paused
    _lib.profile_enable(2)
    K.affine_relu_bwd(x, x, g, b, True)
    _lib.profile_enable(0)
    prof = _lib.profile_read()
    assert prof['affine_fwd'][1] == 3 and prof['affine_bwd'][1] == 1 and prof['reduce_partials'][1] == 1
    assert 0.0 < prof['affine_fwd'][0] < 5.0                        # milliseconds for three tiny kernels
    assert _lib.profile_read_bytes()['affine_fwd'] == 3 * 8.0 * x.numel()
    _lib.profile_enable(1)
    _lib.profile_enable(0)
    assert _lib.profile_read()['affine_fwd'] == (0.0, 0)


@pytest.mark.parametrize('shape', [(128, 512, 4, 4), (64, 64, 32, 32), (3, 5, 7, 9), (1, 1, 1, 263)])
def test_add_relu_fused_tail(K, shape):
    """out = relu(a + b) and d = dy * [out > 0]: bit-exact against the ATen ops they replace."""
    rs = np.random.RandomState(sum(shape))
    a, b, dy = [dev(rs.standard_normal(shape)) for _ in range(3)]
    out = K.add_relu_fwd(a, b)
    assert torch.equal(out, torch.relu(a + b))
    d = K.relu_bwd(dy, out)
    assert torch.equal(d, torch.where(out > 0, dy, torch.zeros_like(dy)))
    from deepipr_amd import passport_ops as P
    big = shape if a.numel() >= P.ADD_RELU_MIN_ELEMENTS else None
    if big:
        a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        (P.add_relu(a1, b1) * dy).sum().backward()
        a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        (torch.relu(a2 + b2) * dy).sum().backward()
        assert torch.equal(a1.grad, a2.grad) and torch.equal(b1.grad, b2.grad)
        # the forked form: two handles of the output, their gradients summed inside deepipr_relu_bwd2
        dy2 = dev(rs.standard_normal(shape))
        assert torch.equal(K.relu_bwd(dy, out, dy2), torch.where(out > 0, dy + dy2, torch.zeros_like(dy)))
        a3, b3 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        o1, o2 = P.add_relu_fork(a3, b3)
        assert torch.equal(o1, o2) and o1.data_ptr() == o2.data_ptr()
        ((o1 * dy).sum() + (o2 * o2 * dy2).sum()).backward()
        a4, b4 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        o = torch.relu(a4 + b4)
        ((o * dy).sum() + (o * o * dy2).sum()).backward()
        assert torch.equal(a3.grad, a4.grad) and torch.equal(b3.grad, b4.grad)
        a5 = a.clone().requires_grad_(True)
        o1, o2 = P.add_relu_fork(a5, b)                      # only one consumer contributes a gradient
        (o2 * dy).sum().backward()
        assert torch.equal(a5.grad, a2.grad)


def _graph_replay_exchange_worker(port):
    """Body of test_graph_replay_with_eager_gradient_exchange; runs in a process of its own (see there)."""
    import sys
    import torch.distributed as dist
    from deepipr_amd.experiments.graph_step import GraphedTrainStep
    from deepipr_amd.experiments.trainer import train_step_v1
    from deepipr_amd.flat_sgd import FlatSGD
    from deepipr_amd import passport_ops
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:%d' % port, rank=0, world_size=1)
    bench, det = torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic
    torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True
    try:
        finals = []
        for graphed in (False, True, 'staged', 'staged-exclusive'):
            print('variant %r: building' % (graphed,), file=sys.stderr, flush=True)
            prod, _ref, x, y = _fullsize_pair(False, 32, 10)
            x, y = x.to(DEV), y.to(DEV)
            opt = FlatSGD(prod.parameters(), **SGD)
            assert opt.comm and len(opt._buckets) >= 3
            if graphed in ('staged', 'staged-exclusive'):
                # the default with several GPUs (experiments/staged.py): the backward stages in ONE graph with an external
                # event behind each, the bucket all-reduces enqueued on a side stream behind those events; "exclusive":
                # the graph is split in front of the stage with split-channel kernels, which waits for the collectives
                from deepipr_amd.experiments.staged import StagedStep
                g = StagedStep(train_step_v1, prod, opt, x, y, graph=True, warmup=0,
                               overlap_sync=graphed == 'staged')
                plan = g.describe()
                assert plan['policy'] == ('shared' if graphed == 'staged' else 'exclusive')
                assert plan['graphs_per_step'] == (1 if graphed == 'staged' else 2), plan
                assert sorted(g._events) == ([0, 1] if graphed == 'staged' else [0])
                assert [s['cut'] for s in plan['stages']] == ['layer4.0', 'layer3.0', None], plan
                # (stage 0 holds the forward pass, whose stem / layer1 / layer2 launches are split-channel -- harmless:
                # nothing is in flight before it; of the backward stages only the last one has them)
                assert [s['split_channel_kernels'] for s in plan['stages']][1:] == [False, True], plan
                assert opt._mode == 'staged'
                for i in range(3):
                    g(x if i % 2 == 0 else x.flip(0), y if i % 2 == 0 else y.flip(0))
            elif graphed:
                g = GraphedTrainStep(train_step_v1, prod, opt, x, y, warmup=0, optimizer_in_graph=False)
                for i in range(3):
                    g(x if i % 2 == 0 else x.flip(0), y if i % 2 == 0 else y.flip(0))
            else:
                for i in range(3):
                    train_step_v1(prod, opt, x if i % 2 == 0 else x.flip(0), y if i % 2 == 0 else y.flip(0))
            torch.cuda.synchronize()
            print('variant %r: three steps done' % (graphed,), file=sys.stderr, flush=True)
            finals.append({k: v.clone() for k, v in prod.state_dict().items()})
        for i, name in ((1, 'one graph + exchange after it'), (2, 'staged, shared'), (3, 'staged, exclusive')):
            states_close(finals[0], finals[i], what='eager vs ' + name)
        passport_ops.kernels.check_exchange()
    finally:
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = bench, det


@pytest.mark.miopen_pinned
def test_graph_replay_with_eager_gradient_exchange(tmp_path):
    """Data-parallel form of the graphed step: forward+backward replayed from a hipGraph, FlatSGD's bucketed
    exchange (forced on in a one-rank nccl group) and the fused SGD kernel run eagerly after each replay.  Must
    equal the plain eager trajectory.

    The body (_graph_replay_exchange_worker) runs in a child process that leaves through os._exit once it has written
    its verdict.  What this guards against: RCCL's watchdog thread polls the end event of every unretired collective;
    a poll that falls inside a stream capture comes back as hipErrorCapturedEvent and the watchdog terminates the process
    (one fresh process in five before distributed.retire_collectives() was put in front of every capture; in the
    long-lived pytest process it showed once in eleven sessions, as a SIGABRT at destroy_process_group:
    profiles/r04_pytest_gpu_7_crash.log, r04_nccl_flake_probe.txt).  The product now keeps such polls out of its captures;
    a library thread that can still abort the interpreter must not be able to take the session with it."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    out = tmp_path / 'verdict.json'
    env = dict(os.environ, DEEPIPR_FORCE_DDP='1')           # (the pinned MIOpen environment of conftest.py is inherited)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, '-m', 'tests.test_parity_gpu', str(port), str(out)], cwd=root, env=env,
                          capture_output=True, text=True, timeout=900)
    assert out.exists(), 'the worker died before its verdict (rc %s):\n%s\n%s' % (proc.returncode, proc.stdout[-3000:], proc.stderr[-3000:])
    verdict = json.loads(out.read_text())
    assert verdict['ok'], verdict['error']


def test_product_has_no_cpu_path():
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    blk = PassportBlock(4, 16, 3, 1, 1, {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1})
    with pytest.raises(RuntimeError, match='GPU only'):
        blk(torch.randn(2, 4, 8, 8))


def test_weight_shuttle_with_key_derived_gamma_beta(golden_dir):
    """load_passport_model_to_normal_model (experiments/utils.py:191-239) on a V1 net without learnable scale/bias:
    the plain net's norm weights become gamma/beta computed from the keys by the HIP GEMV."""
    from tests.impls import ProductShuttle
    want = load_golden(golden_dir, 'shuttle')
    got = runner.collect_shuttle(ProductShuttle('cuda:0'), with_keys=True)
    keyed = [k for k in got if '_p2n_keys/' in k]
    assert len(keyed) == 2 * (3 + 5)
    for k in sorted(got):
        if '_p2n_keys/' in k:
            scale = float(np.abs(want[k]).max()) + 1e-12
            assert np.abs(got[k] - want[k]).max() <= 1e-5 * max(1.0, scale), k
            if k.endswith('bn.weight'):
                assert np.array_equal(np.sign(got[k]), np.sign(want[k])), k
        else:
            assert np.array_equal(got[k], want[k]), k


def test_force_passport_paths_on_gpu(golden_dir):
    """flip_attack.py:25 / pruning_attack.py:26 / passportconv2d.py:142-175: with the learnable pair installed
    (init_scale/init_bias(True), what the weight shuttles do) plain calls use it, force_passport=True returns to the
    key-derived gamma/beta and refreshes the sign loss; private blocks: force_passport overrides ind=0."""
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
    gold = load_golden(golden_dir, 'blocks')
    kw = {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}
    x = torch.from_numpy(gold['ckpt_in/x']).to(DEV)
    for fuse in (True, False):
        blk = PassportBlock(4, 16, 3, 1, 1, kw)
        blk.fuse_norm = fuse
        blk.load_state_dict({k[len('ckpt_v1/'):]: torch.from_numpy(np.array(v)) for k, v in gold.items() if k.startswith('ckpt_v1/')}, strict=True)
        blk = blk.to(DEV).eval()
        blk.init_scale(True)
        blk.init_bias(True)
        assert blk.scale.device == x.device and blk.scale.requires_grad
        with torch.no_grad():
            blk.scale.copy_(torch.from_numpy(gold['force/scale']))
            blk.bias.copy_(torch.from_numpy(gold['force/bias']))
            close(blk(x).cpu().numpy(), gold['force/v1_plain'], 'learnable pair', 1e-4, 1e-5)
            assert np.array_equal(blk.get_scale().cpu().numpy().reshape(-1), gold['force/scale'])
            blk.sign_loss.reset()
            close(blk(x, force_passport=True).cpu().numpy(), gold['force/v1_forced'], 'forced', 1e-4, 1e-5)
            g = blk.get_scale(True).cpu().numpy().reshape(-1)
            close(g, gold['force/v1_forced_scale'], 'forced gamma', 1e-5, 1e-6)
            assert np.array_equal(np.sign(g), np.sign(gold['force/v1_forced_scale']))
            close(blk.get_bias(True).cpu().numpy().reshape(-1), gold['force/v1_forced_bias'], 'forced beta', 1e-5, 1e-6)
            close(np.float64(float(blk.sign_loss.loss)), gold['force/v1_forced_sign_loss'], 'sign loss', 1e-5, 1e-6)
            assert float(blk.sign_loss.acc) == float(gold['force/v1_forced_sign_acc'])
    pv = PassportPrivateBlock(4, 16, 3, 1, 1, kw)
    pv.load_state_dict({k[len('ckpt_private/'):]: torch.from_numpy(np.array(v)) for k, v in gold.items() if k.startswith('ckpt_private/')}, strict=True)
    pv = pv.to(DEV).eval()
    with torch.no_grad():
        close(pv(x, force_passport=True, ind=0).cpu().numpy(), gold['force/private_forced_ind0'],
              'private forced', 1e-4, 1e-5)


if __name__ == '__main__':                                   # the child process of test_graph_replay_with_eager_gradient_exchange
    import json
    import sys
    import traceback
    verdict = {'ok': True, 'error': None}
    try:
        _graph_replay_exchange_worker(int(sys.argv[1]))
    except BaseException:                                    # noqa: B036 -- the parent re-raises it as an assertion
        verdict = {'ok': False, 'error': traceback.format_exc()}
    with open(sys.argv[2], 'w') as f:
        json.dump(verdict, f)
        f.flush()
        os.fsync(f.fileno())
    os._exit(0)                                              # no communicator teardown, no atexit handlers
