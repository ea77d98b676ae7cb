"""deepipr_conv_fwd / deepipr_conv_dgrad -- the data convolution and its backward-data on the fp32 matrix cores -- against
the oracle: ATen's convolution / convolution_backward evaluated in float64 (what `self.conv(x)`,
models/layers/passportconv2d.py:218 / models/layers/conv2d.py:31, and its autograd backward compute in the reference).
Bar: 1e-5 of the result's scale (fp32 accumulation over up to 4 608 products; measured 2e-7 - 1.2e-6), bit-reproducible."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def K():
    """This module pins the DIRECT implicit-GEMM family (the stride-1 3x3 instances are what every shape outside the Winograd
    kernels' reach takes, and what DEEPIPR_CONV_ALGO=direct selects); tests/test_conv_wino_gpu.py covers the default."""
    from deepipr_amd.passport_ops import kernels
    assert torch.cuda.is_available(), 'needs an MI355X'
    before = kernels.set_conv_algo('direct')
    yield kernels
    kernels.set_conv_algo(before)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _conv64(x, w, st, pad):
    return torch.ops.aten.convolution(x.double(), w.double(), None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1)


def _dgrad64(dy, x, w, st, pad):
    return torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [st, st], [pad, pad], [1, 1], False,
                                               [0, 0], 1, [True, False, False])[0]


# (N, Ci, Co, H, W of the input, k, stride): every instance of the kernel family -- large / small position counts
# (row-band vs k-group tiles), Ci != Co, the config R and config P shard shapes of ResNet18
SHAPES = [
    (128, 64, 64, 32, 32, 3, 1), (8, 64, 128, 32, 32, 3, 1), (128, 128, 128, 16, 16, 3, 1), (4, 128, 64, 16, 16, 3, 1),
    (128, 256, 256, 8, 8, 3, 1), (3, 256, 128, 8, 8, 3, 1), (128, 512, 512, 4, 4, 3, 1), (4, 192, 64, 4, 4, 3, 1),
    (32, 64, 64, 32, 32, 3, 1), (32, 512, 512, 4, 4, 3, 1), (32, 256, 256, 8, 8, 3, 1), (8, 512, 512, 4, 4, 3, 1),
    (4, 512, 256, 4, 4, 3, 1), (16, 256, 128, 8, 8, 3, 1),
    (128, 64, 128, 32, 32, 3, 2), (3, 64, 64, 32, 32, 3, 2), (128, 128, 256, 16, 16, 3, 2), (5, 128, 64, 16, 16, 3, 2),
    (128, 256, 512, 8, 8, 3, 2), (4, 256, 64, 8, 8, 3, 2), (32, 256, 512, 8, 8, 3, 2),
    (128, 64, 128, 32, 32, 1, 2), (2, 64, 64, 32, 32, 1, 2), (128, 128, 256, 16, 16, 1, 2), (128, 256, 512, 8, 8, 1, 2),
    (4, 256, 64, 8, 8, 1, 2), (2, 64, 64, 16, 32, 3, 1), (2, 64, 64, 16, 32, 3, 2),
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_forward_and_backward_data_match_the_float64_oracle(K, shape):
    n, ci, co, h, w, k, st = shape
    pad = k // 2
    x, wt = _rand((n, ci, h, w), 1 + n), _rand((co, ci, k, k), 2 + co, 0.05)
    dy = _rand((n, co, h // st, w // st), 3 + ci)
    y = K.conv_fwd(x, wt, st, pad)
    assert y is not None and y.shape == (n, co, h // st, w // st)
    ref = _conv64(x, wt, st, pad)
    assert float((y.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(y, K.conv_fwd(x, wt, st, pad))
    dx = K.conv_dgrad(dy, wt, x.shape, st, pad)
    assert dx is not None and dx.shape == x.shape
    ref = _dgrad64(dy, x, wt, st, pad)
    assert float((dx.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(dx, K.conv_dgrad(dy, wt, x.shape, st, pad))


@pytest.mark.parametrize('shape', [(32, 512, 512, 4, 4, 3, 1), (128, 512, 512, 4, 4, 3, 1), (32, 256, 256, 8, 8, 3, 1),
                                   (32, 256, 512, 8, 8, 3, 2), (8, 320, 192, 4, 4, 3, 1)], ids=lambda s: 'x'.join(map(str, s)))
def test_split_k_form_agrees_with_the_plain_form(K, shape):
    """Few output positions (deep layers, small batches): K = Cin * 9 is split over workgroups, partial outputs summed by
    a second launch in split order (deepipr_conv_fwd_ws / _dgrad_ws).  Same result as the plain form -- the ABI v7 entry
    points without a workspace -- to rounding (a different association of the same sum), both against float64, both
    bit-reproducible; the planner really splits these shapes."""
    from deepipr_amd import _lib
    n, ci, co, h, w, k, st = shape
    pad = k // 2
    x, wt = _rand((n, ci, h, w), 21 + n), _rand((co, ci, k, k), 22 + co, 0.05)
    dy = _rand((n, co, h // st, w // st), 23 + ci)
    assert K.conv_workspace(n, ci, co, h, w, k, st, pad, 0) > 0
    y = K.conv_fwd(x, wt, st, pad)
    plain = torch.empty_like(y)
    _lib.check(_lib.lib().deepipr_conv_fwd(x.data_ptr(), wt.data_ptr(), plain.data_ptr(), n, ci, co, h, w, k, st, pad, None),
               'conv_fwd')
    ref = _conv64(x, wt, st, pad)
    scale = float(ref.abs().max())
    assert not torch.equal(y, plain)
    assert float((y - plain).abs().max()) <= 2e-6 * scale
    assert float((y.double() - ref).abs().max()) <= 1e-5 * scale and float((plain.double() - ref).abs().max()) <= 1e-5 * scale
    assert torch.equal(y, K.conv_fwd(x, wt, st, pad))
    if st == 1:                                             # stride-2 backward-data never splits
        assert K.conv_workspace(n, ci, co, h, w, k, st, pad, 1) > 0
        dx = K.conv_dgrad(dy, wt, x.shape, st, pad)
        plain = torch.empty_like(dx)
        _lib.check(_lib.lib().deepipr_conv_dgrad(dy.data_ptr(), wt.data_ptr(), plain.data_ptr(), n, ci, co, h, w, k, st, pad,
                                                 None), 'conv_dgrad')
        ref = _dgrad64(dy, x, wt, st, pad)
        scale = float(ref.abs().max())
        assert float((dx - plain).abs().max()) <= 2e-6 * scale and float((dx.double() - ref).abs().max()) <= 1e-5 * scale
        assert torch.equal(dx, K.conv_dgrad(dy, wt, x.shape, st, pad))
    else:
        assert K.conv_workspace(n, ci, co, h, w, k, st, pad, 1) == 0


@pytest.mark.parametrize('k,st', [(3, 1), (3, 2), (1, 2)])
def test_one_hot_operands_are_exact(K, k, st):
    """Small-integer one-hot inputs: every output is a single product (or a short exact sum), so a wrong tap, a halo that
    is not zero, a parity class written to the wrong pixel or a band leaking into its neighbour is an exact mismatch."""
    n, c, h, pad = 4, 64, 8 * st, k // 2
    rs = np.random.RandomState(k + st)
    x = torch.zeros(n, c, h, h, device=DEV)
    dy = torch.zeros(n, c, h // st, h // st, device=DEV)
    w = torch.zeros(c, c, k, k, device=DEV)
    for _ in range(200):
        x[rs.randint(n), rs.randint(c), rs.choice([0, h - 1, rs.randint(h)]), rs.choice([0, h - 1, rs.randint(h)])] = float(rs.randint(1, 5))
        dy[rs.randint(n), rs.randint(c), rs.randint(h // st), rs.randint(h // st)] = float(rs.randint(1, 5))
    for _ in range(600):
        w[rs.randint(c), rs.randint(c), rs.randint(k), rs.randint(k)] = float(rs.randint(1, 4))
    y, dx = K.conv_fwd(x, w, st, pad), K.conv_dgrad(dy, w, x.shape, st, pad)
    ry, rdx = _conv64(x, w, st, pad), _dgrad64(dy, x, w, st, pad)
    assert float(ry.abs().sum()) > 0 and float(rdx.abs().sum()) > 0
    assert torch.equal(y.double(), ry) and torch.equal(dx.double(), rdx)


@pytest.mark.parametrize('n,h', [(5, 224), (2, 64), (1, 8)])
def test_imagenet_stem_forward_matches_the_float64_oracle(K, n, h):
    """k_conv_stem7_fwd (deepipr_conv_stem7.inc): Conv 3 -> 64, 7x7, stride 2, pad 3 on 224-wide images
    (models/resnet_passport.py:94-98), any height that is a multiple of 8.  1e-5 of scale against float64 (147 products per
    output), bit-reproducible; the band borders (first / last rows of an image, the halo rows between bands) are where a wrong
    row offset shows, so the error is also taken per output row."""
    assert K.conv_supported(n, 3, 64, h, 224, 7, 2, 3, 0) and not K.conv_supported(n, 3, 64, h, 224, 7, 2, 3, 1)
    x, wt = _rand((n, 3, h, 224), 31 + n), _rand((64, 3, 7, 7), 32 + h, 0.1)
    y = K.conv_fwd(x, wt, 2, 3)
    assert y is not None and y.shape == (n, 64, h // 2, 112)
    ref = _conv64(x, wt, 2, 3)
    scale = float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) <= 1e-5 * scale
    per_row = (y.double() - ref).abs().amax(dim=(0, 1, 3))
    assert float(per_row.max()) <= 1e-5 * scale, per_row
    assert torch.equal(y, K.conv_fwd(x, wt, 2, 3))


def test_imagenet_stem_forward_is_exact_on_small_integers(K):
    """Every tap, every border: small-integer images and filters (all 147 taps non-zero, values in -3 .. 3) give sums below 2^24 --
    the fp32 result must EQUAL the float64 one, in particular in the first / last three rows and columns (the zero padding) and
    in the padded eighth column of the kernel's K layout (a non-zero there would show as an exact mismatch)."""
    rs = np.random.RandomState(7)
    n, h = 3, 32
    x = torch.from_numpy(rs.randint(-3, 4, size=(n, 3, h, 224)).astype(np.float32)).to(DEV)
    w = torch.from_numpy(rs.randint(-3, 4, size=(64, 3, 7, 7)).astype(np.float32)).to(DEV)
    y, ref = K.conv_fwd(x, w, 2, 3), _conv64(x, w, 2, 3)
    assert float(ref.abs().max()) > 50
    assert torch.equal(y.double(), ref)
    # one-hot filters: output = one shifted, subsampled image plane
    w1 = torch.zeros(64, 3, 7, 7, device=DEV)
    for m in range(64):
        w1[m, m % 3, (m * 5) % 7, (m * 3) % 7] = 1.0
    assert torch.equal(K.conv_fwd(x, w1, 2, 3).double(), _conv64(x, w1, 2, 3))


@pytest.mark.parametrize('n,h', [(128, 32), (5, 32), (2, 8), (3, 64)])
def test_cifar_stem_forward_matches_the_float64_oracle_and_is_exact_on_small_integers(K, n, h):
    """k_conv_stem3_fwd: Conv 3 -> 64, 3x3, stride 1, pad 1 on 32-wide images (models/resnet_passport.py:99-101) -- the last
    convolution of the headline configuration that ran in the vendor library.  1e-5 of scale against float64, bit-reproducible;
    small integers exact (every tap, the zero padding, the padded fourth kernel column)."""
    assert K.conv_supported(n, 3, 64, h, 32, 3, 1, 1, 0) and not K.conv_supported(n, 3, 64, h, 32, 3, 1, 1, 1)
    x, wt = _rand((n, 3, h, 32), 51 + n), _rand((64, 3, 3, 3), 52 + h, 0.2)
    y = K.conv_fwd(x, wt, 1, 1)
    assert y is not None and y.shape == (n, 64, h, 32)
    ref = _conv64(x, wt, 1, 1)
    assert float((y.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(y, K.conv_fwd(x, wt, 1, 1))
    rs = np.random.RandomState(n + h)
    xi = torch.from_numpy(rs.randint(-4, 5, size=(n, 3, h, 32)).astype(np.float32)).to(DEV)
    wi = torch.from_numpy(rs.randint(-4, 5, size=(64, 3, 3, 3)).astype(np.float32)).to(DEV)
    assert torch.equal(K.conv_fwd(xi, wi, 1, 1).double(), _conv64(xi, wi, 1, 1))


@pytest.mark.parametrize('case', [dict(ci=3, co=64, k=7, st=2, pad=3, h=56, wd=112), dict(ci=3, co=32, k=7, st=2, pad=3, h=224, wd=224),
                                  dict(ci=3, co=64, k=7, st=2, pad=3, h=12, wd=224), dict(ci=3, co=64, k=7, st=2, pad=2, h=224, wd=224),
                                  dict(ci=4, co=64, k=7, st=2, pad=3, h=224, wd=224)])
def test_stem_shapes_outside_the_kernel_are_refused(K, case):
    assert not K.conv_supported(2, case['ci'], case['co'], case['h'], case['wd'], case['k'], case['st'], case['pad'], 0)
    x, w = _rand((2, case['ci'], case['h'], case['wd']), 1), _rand((case['co'], case['ci'], case['k'], case['k']), 2)
    assert K.conv_fwd(x, w, case['st'], case['pad']) is None


@pytest.mark.parametrize('case', [dict(ci=3, co=64), dict(ci=64, co=96), dict(h=14), dict(k=5, pad=2), dict(k=1, pad=0, st=1, ci=32), dict(k=1, pad=1, st=1),
                                  dict(h=64, st=2), dict(n=3, h=4), dict(k=3, pad=0)])
def test_shapes_outside_the_kernels_are_refused(K, case):
    n, ci, co, h = case.get('n', 4), case.get('ci', 64), case.get('co', 64), case.get('h', 8)
    k, st = case.get('k', 3), case.get('st', 1)
    pad = case.get('pad', k // 2)
    assert not K.conv_supported(n, ci, co, h, h, k, st, pad, 0)
    x, w = _rand((n, ci, h, h), 1), _rand((co, ci, k, k), 2)
    assert K.conv_fwd(x, w, st, pad) is None
    if not (case.get('h') == 64):                           # the 3x3 stride-2 backward-data also takes 32-wide dy maps
        assert not K.conv_supported(n, ci, co, h, h, k, st, pad, 1)


def test_model_level_own_convolutions_equal_the_library_path(K, monkeypatch):
    """A residual block with a projection (stride-2 3x3 + stride-1 3x3 + 1x1 stride-2 shortcut) through DEEPIPR_OWN_CONV =
    all / auto / 0: same outputs and gradients within 2e-5 of scale (different summation orders), and the kernels that
    ran are the ones the policy names."""
    from deepipr_amd import _lib
    from deepipr_amd import passport_ops as P
    from deepipr_amd.models.resnet_passport import BasicPassportBlock
    K.set_conv_algo('winograd')                                 # the library default (the fixture restores its own setting)
    kw = {name: {'flag': False, 'norm_type': 'bn'} for name in ('convbnrelu_1', 'convbn_2', 'shortcut')}
    res = {}
    for mode in ('all', 'auto', '0'):
        monkeypatch.setattr(P, 'OWN_CONV', mode)
        monkeypatch.setattr(P, 'OWN_WGRAD', mode != '0')
        torch.manual_seed(0)
        blk = BasicPassportBlock(64, 128, 2, kw).to(DEV)
        x = _rand((32, 64, 32, 32), 9).requires_grad_(True)
        _lib.profile_enable(True)
        y = blk(x)
        y.square().mean().backward()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        prof = _lib.profile_read()
        res[mode] = (y.detach(), x.grad, [p.grad for p in blk.parameters()],
                     (int(prof['conv_fwd'][1]) + int(prof['conv_wino_fwd'][1]),          # direct + Winograd instances
                      int(prof['conv_dgrad'][1]) + int(prof['conv_wino_dgrad'][1]),
                      int(prof['conv_wgrad'][1]) + int(prof['conv_wgrad_b3'][1]) + int(prof['conv_wino_wgrad'][1])))     # fp32-MFMA direct + bf16x3 + Winograd instances
    # auto: the stride-1 3x3 takes the Winograd kernel at any size (round 5; the direct kernel needed >= 32 768 positions)
    assert res['all'][3] == (3, 3, 3) and res['auto'][3] == (3, 3, 3) and res['0'][3] == (0, 0, 0)
    K.set_conv_algo('direct')
    for mode in ('all', 'auto'):
        for a, b in zip([res[mode][0], res[mode][1]] + res[mode][2], [res['0'][0], res['0'][1]] + res['0'][2]):
            assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-9
