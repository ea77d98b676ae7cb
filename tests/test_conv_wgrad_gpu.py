"""deepipr_conv_wgrad -- the data convolution's weight gradient on the matrix cores -- against the oracle.

The reference op is the weight half of the autograd backward of `self.conv(x)` (models/layers/passportconv2d.py:218,
models/layers/conv2d.py:31), i.e. ATen's convolution_backward; the oracle here is that same ATen op evaluated in
float64 (oracle/torch_ref.py composes its layers from it).  Bar: 1e-5 of the gradient's scale (fp32 accumulation over up
to 131 072 products; measured 3-6e-7, the vendor library's own fp32 result sits at 4e-7 - 1e-6), bit-reproducible.
Every test runs in both arithmetic modes of the 3x3 stride-1 instances: 'fp32' (the default: the fp32 MFMA) and 'bf16x3'
(opt-in: fp32 operands split exactly into three bf16 words, six products on the bf16 matrix cores, fp32 accumulation);
the shapes the bf16x3 kernel does not cover take the fp32 MFMA either way."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module', params=['winograd', 'fp32', 'bf16x3'])
def K(request):
    """'winograd': the default -- the 3x3 stride-1 instances run Winograd F(3x3, 2x2) on the fp32 MFMA (round 5);
    'fp32': the direct fp32-MFMA kernel (DEEPIPR_CONV_ALGO=direct); 'bf16x3': the opt-in split-bf16 arithmetic."""
    from deepipr_amd.passport_ops import kernels
    assert torch.cuda.is_available(), 'needs an MI355X'
    algo = kernels.set_conv_algo('winograd' if request.param == 'winograd' else 'direct')
    before = kernels.set_conv_arith('bf16x3' if request.param == 'bf16x3' else 'fp32')
    yield kernels
    kernels.set_conv_arith(before)
    kernels.set_conv_algo(algo)


def _ref(x, dy, wshape, stride=1, pad=1):
    # (the oracle: ATen's own convolution_backward in float64)
    w = torch.zeros(wshape, dtype=torch.float64, device=x.device)
    return torch.ops.aten.convolution_backward(dy.double(), x.double(), w, None, [stride, stride], [pad, pad], [1, 1],
                                               False, [0, 0], 1, [False, True, False])[1]


def _rand(shape, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randn(shape, generator=g).to(DEV)


# (N, Ci, Co, H, W): every map width of the kernel; one tile / many tiles; Ci != Co; batch sizes that leave the last
# split ragged (chunks not a multiple of the split count); the config R and config P shard shapes of ResNet18's layers
SHAPES = [
    (128, 64, 64, 32, 32), (32, 64, 64, 32, 32), (3, 64, 64, 32, 32),
    (128, 128, 128, 16, 16), (5, 64, 128, 16, 16), (32, 128, 128, 16, 16),
    (128, 256, 256, 8, 8), (7, 256, 128, 8, 8), (32, 256, 256, 8, 8),
    (128, 512, 512, 4, 4), (2, 512, 512, 4, 4), (32, 512, 512, 4, 4), (6, 128, 192, 4, 4),
    (2, 64, 64, 8, 32), (4, 64, 64, 16, 4),
    # stride 2 (the first convolution of layer2 / 3 / 4): (N, Ci, Co, H, W of the INPUT, 2)
    (128, 64, 128, 32, 32, 2), (5, 64, 64, 32, 32, 2), (128, 128, 256, 16, 16, 2), (3, 128, 64, 16, 16, 2),
    (128, 256, 512, 8, 8, 2), (1, 256, 64, 8, 8, 2), (2, 64, 64, 16, 32, 2),
    # 1x1 stride 2 pad 0 (the projection shortcuts): (N, Ci, Co, H, W of the INPUT, 2, 1)
    (128, 64, 128, 32, 32, 2, 1), (3, 64, 64, 32, 32, 2, 1), (128, 128, 256, 16, 16, 2, 1), (5, 128, 64, 16, 16, 2, 1),
    (128, 256, 512, 8, 8, 2, 1), (2, 256, 64, 8, 8, 2, 1), (32, 256, 512, 8, 8, 2, 1),
    # (the Winograd instances also take Ci a multiple of 32 and any batch on 4-wide maps: test_winograd_only_shapes)
    # 1x1 stride 1 pad 0 (the Bottleneck's convbnrelu_1 / convbn_3 / stride-1 projection, BASELINE config 5;
    # deepipr_conv_1x1.inc): every chunk geometry (64 | 56 | 2 x 28 positions, the 49-position plane with 4-byte loads),
    # every workgroup tile (64 / 128 channels either side), odd batches on the two-image chunks, ragged last splits
    (4, 64, 64, 56, 56, 1, 1), (3, 64, 256, 56, 56, 1, 1), (2, 256, 64, 56, 56, 1, 1), (2, 256, 128, 56, 56, 1, 1),
    (32, 256, 64, 56, 56, 1, 1),
    (5, 128, 512, 28, 28, 1, 1), (3, 192, 64, 28, 28, 1, 1), (2, 64, 128, 28, 28, 1, 1), (3, 512, 128, 28, 28, 1, 1),
    (5, 256, 1024, 14, 14, 1, 1), (4, 1024, 256, 14, 14, 1, 1), (1, 64, 128, 14, 14, 1, 1), (7, 128, 64, 14, 14, 1, 1),
    (7, 512, 2048, 7, 7, 1, 1), (3, 2048, 512, 7, 7, 1, 1), (2, 64, 64, 7, 7, 1, 1), (5, 64, 128, 7, 7, 1, 1),
    (2, 128, 64, 7, 7, 1, 1), (256, 512, 2048, 7, 7, 1, 1), (64, 1024, 256, 14, 14, 1, 1),
    (2, 64, 64, 8, 8, 1, 1), (3, 128, 128, 16, 4, 1, 1),
    # the stem: 3 input channels, (ci, tap) = 27 columns of one accumulator tile
    (128, 3, 64, 32, 32), (5, 3, 64, 32, 32), (32, 3, 128, 32, 32), (2, 3, 64, 8, 32),
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_wgrad_matches_the_float64_oracle_and_is_bit_reproducible(K, shape):
    n, ci, co, h, w = shape[:5]
    st = shape[5] if len(shape) > 5 else 1
    k = shape[6] if len(shape) > 6 else 3
    pad = k // 2
    x, dy = _rand((n, ci, h, w), 1 + n + ci), _rand((n, co, h // st, w // st), 2 + n + co)
    got = K.conv_wgrad(x, dy, (co, ci, k, k), st, pad)
    assert got is not None and got.shape == (co, ci, k, k) and got.is_contiguous()
    ref = _ref(x, dy, (co, ci, k, k), st, pad)
    scale = float(ref.abs().max())
    assert float((got.double() - ref).abs().max()) <= 1e-5 * scale
    again = K.conv_wgrad(x, dy, (co, ci, k, k), st, pad)
    assert torch.equal(got, again)


def test_two_k_ranges_per_workgroup_agree_with_one(tmp_path):
    """k_conv_wino_wgrad2 runs two K ranges in one eight-wavefront workgroup and leaves ONE partial tile per pair (half the bytes
    for k_conv_wgrad_reduce); the planner takes it for short ranges only (<= 4 chunks: it measured slower on config R).  Forced on
    (DEEPIPR_WGRAD_PAIR=2) and off (=0), a process each: both within 1e-5 of scale of float64 ATen, exact on small integers,
    bit-reproducible; the forced runs ask for half the workspace and agree with the plain ones to rounding (a different
    association of the same sum)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    got = {}
    for mode in ('2', '0'):
        out = str(tmp_path / ('pair%s.npz' % mode))
        subprocess.run([sys.executable, os.path.join(here, 'wgrad_pair_case.py'), out], check=True, timeout=600,
                       env=dict(os.environ, DEEPIPR_WGRAD_PAIR=mode, DEEPIPR_CONV_ALGO='winograd'))
        got[mode] = np.load(out)
    from tests.wgrad_pair_case import SHAPES
    halved = 0
    for i in range(len(SHAPES)):
        for mode in ('2', '0'):
            d = got[mode]
            assert float(d['err_%d' % i]) <= 1e-5 and bool(d['repeat_%d' % i]) and bool(d['exact_%d' % i]), (SHAPES[i], mode)
        a, b = int(got['2']['ws_%d' % i]), int(got['0']['ws_%d' % i])
        assert 0 < a <= b
        halved += int(2 * a <= b + b // 8)
        x, y = got['2']['dw_%d' % i], got['0']['dw_%d' % i]
        assert np.abs(x - y).max() <= 2e-6 * np.abs(y).max()
    assert halved >= len(SHAPES) - 2                           # (a single K range has nothing to pair)


@pytest.mark.parametrize('n,h', [(5, 224), (3, 64), (1, 8), (2, 226)])
def test_imagenet_stem_wgrad_matches_the_float64_oracle(n, h):
    """k_conv_stem7_wgrad (deepipr_conv_stem7.inc): the weight gradient of Conv 3 -> 64, 7x7, stride 2, pad 3 on 224-wide images
    (models/resnet_passport.py:94-98).  K = N * H / 2 * 112 output positions summed in fp32 (up to 62 720 here; 3.2 M at batch
    256): 1e-5 of scale against float64, bit-reproducible (fixed order: wavefronts in LDS, workgroups in the reduce launch)."""
    from deepipr_amd.passport_ops import kernels as K
    assert K.conv_wgrad_workspace(n, 3, 64, h, 224, 7, 7, 2, 3) > 0
    x, dy = _rand((n, 3, h, 224), 41 + n), _rand((n, 64, h // 2, 112), 42 + h)
    got = K.conv_wgrad(x, dy, (64, 3, 7, 7), 2, 3)
    assert got is not None and got.shape == (64, 3, 7, 7) and got.is_contiguous()
    ref = _ref(x, dy, (64, 3, 7, 7), 2, 3)
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(got, K.conv_wgrad(x, dy, (64, 3, 7, 7), 2, 3))


def test_imagenet_stem_wgrad_is_exact_on_small_integers_and_sees_the_padding():
    """Small-integer x and dy (sums below 2^24): the fp32 result EQUALS the float64 one -- every tap, the three zero rows /
    columns around the image, the idle lanes behind tap 146; then one-hot operands at the image corners."""
    from deepipr_amd.passport_ops import kernels as K
    rs = np.random.RandomState(11)
    n, h = 3, 32
    x = torch.from_numpy(rs.randint(-3, 4, size=(n, 3, h, 224)).astype(np.float32)).to(DEV)
    dy = torch.from_numpy(rs.randint(-2, 3, size=(n, 64, h // 2, 112)).astype(np.float32)).to(DEV)
    got, ref = K.conv_wgrad(x, dy, (64, 3, 7, 7), 2, 3), _ref(x, dy, (64, 3, 7, 7), 2, 3)
    assert float(ref.abs().max()) > 100
    assert torch.equal(got.double(), ref)
    x1, dy1 = torch.zeros_like(x), torch.zeros_like(dy)
    for (a, b) in [(0, 0), (0, 223), (h - 1, 0), (h - 1, 223), (5, 100)]:
        x1[rs.randint(n), rs.randint(3), a, b] = float(rs.randint(1, 5))
    for (a, b) in [(0, 0), (0, 111), (h // 2 - 1, 0), (h // 2 - 1, 111), (2, 50)]:
        dy1[:, rs.randint(64), a, b] = float(rs.randint(1, 5))
    assert torch.equal(K.conv_wgrad(x1, dy1, (64, 3, 7, 7), 2, 3).double(), _ref(x1, dy1, (64, 3, 7, 7), 2, 3))


@pytest.mark.parametrize('st', [1, 2])
def test_wgrad_sees_the_zero_padding_and_every_tap(K, st):
    """One-hot dy / x: every dW entry is a single product, so a wrong tap offset, a halo column that is not zero or a
    row band that leaks into its neighbour shows as an exact mismatch."""
    n, c, h = 2, 64, 8 * st
    ho = h // st
    x = torch.zeros(n, c, h, h, device=DEV)
    dy = torch.zeros(n, c, ho, ho, device=DEV)
    rs = np.random.RandomState(0)
    for _ in range(60):
        x[rs.randint(n), rs.randint(c), rs.choice([0, h - 1, rs.randint(h)]), rs.choice([0, h - 1, rs.randint(h)])] = float(rs.randint(1, 9))
        dy[rs.randint(n), rs.randint(c), rs.choice([0, ho - 1, rs.randint(ho)]), rs.choice([0, ho - 1, rs.randint(ho)])] = float(rs.randint(1, 9))
    got = K.conv_wgrad(x, dy, (c, c, 3, 3), st, 1)
    ref = _ref(x, dy, (c, c, 3, 3), st)
    assert float(ref.abs().sum()) > 0
    assert torch.equal(got.double(), ref)                               # small integers: exact in fp32


@pytest.mark.parametrize('shape', [(128, 64, 64, 32, 32), (128, 128, 128, 16, 16), (128, 256, 256, 8, 8), (32, 64, 128, 16, 16)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_bf16x3_is_as_accurate_as_the_fp32_mfma_and_the_library(K, shape):
    """The claim behind the opt-in arithmetic (include/deepipr_hip.h, "Arithmetic"): against the float64 oracle the
    bf16x3 result carries the error of an fp32 computation -- RMS error within 2x of the fp32-MFMA kernel's (it is a
    different summation order: the K-groups combine differently) and not above 1.5x the vendor library's fp32 result,
    worst element within 1e-6 of scale; operands spread over 22 binary orders of magnitude make no difference (a bf16 word
    has fp32's exponent range)."""
    if K.conv_arith() != 'bf16x3':
        pytest.skip('compares the two modes itself')
    n, ci, co, h, w = shape
    wshape = (co, ci, 3, 3)
    for xs, ds in ((1.0, 1.0), (1e-12, 1e10)):
        x, dy = _rand((n, ci, h, w), 11) * xs, _rand((n, co, h, w), 12) * ds
        ref = _ref(x, dy, wshape)
        scale = float(ref.abs().max())
        rms = lambda a: float(((a.double() - ref) ** 2).mean().sqrt()) / scale
        b3 = K.conv_wgrad(x, dy, wshape, 1, 1)
        K.set_conv_arith('fp32')
        try:
            f32 = K.conv_wgrad(x, dy, wshape, 1, 1)
        finally:
            K.set_conv_arith('bf16x3')
        lib = torch.ops.aten.convolution_backward(dy, x, torch.empty(wshape, device=DEV), None, [1, 1], [1, 1], [1, 1], False,
                                                  [0, 0], 1, [False, True, False])[1]
        assert not torch.equal(b3, f32)                                     # really two different kernels
        assert rms(b3) <= 2.0 * rms(f32) and rms(b3) <= 1.5 * rms(lib), (rms(b3), rms(f32), rms(lib))
        assert float((b3.double() - ref).abs().max()) <= 1e-6 * scale


def test_bf16x3_split_is_exact_for_every_fp32_value(K):
    """x = h + m + l: with a one-hot dy every dW entry is ONE fp32 value of x passed through the three-way split and the
    fp32 accumulator -- any 24-bit significand, any sign, magnitudes from 1e-25 to 1e25, must come back bit for bit
    (below about 2^-110 the third word of the split leaves bf16's normal range: include/deepipr_hip.h)."""
    if K.conv_algo() == 'winograd':
        pytest.skip('a property of the direct kernels: Winograd sums neighbouring values BEFORE it multiplies, so an operand '
                    '50 orders of magnitude below its neighbours is lost (as in the vendor library\'s F(2x3) kernels); its '
                    'exactness test is test_one_hot_small_integers_are_exact_in_the_winograd_kernel')
    n, c, h = 1, 64, 8
    g = torch.Generator(device='cpu').manual_seed(3)
    x = (torch.randn(n, c, h, h, generator=g) * torch.pow(10.0, torch.randint(-25, 26, (n, c, h, h), generator=g).float())).to(DEV)
    dy = torch.zeros(n, c, h, h, device=DEV)
    dy[0, :, 3, 4] = 1.0                                                 # dW[co][ci][r][s] = x[0][ci][3 + r - 1][4 + s - 1]
    got = K.conv_wgrad(x, dy, (c, c, 3, 3), 1, 1)
    want = x[0, :, 2:5, 3:6].unsqueeze(0).expand(c, c, 3, 3)
    assert torch.equal(got, want)


@pytest.mark.parametrize('shape', [(3, 64, 64, 4, 4), (66, 512, 512, 4, 4), (7, 32, 64, 8, 8), (5, 96, 128, 16, 16), (2, 160, 64, 2, 32),
                                   (132, 512, 512, 4, 4), (9, 32, 192, 12, 4),
                                   # ImageNet-geometry maps (BASELINE config 5; k_conv_wino_wgrad_x): two column segments per
                                   # 56-wide row, 8-byte items and two-image chunks on 14-wide maps (odd batches masked), 4-byte
                                   # items and half-phantom tiles on 7 x 7 maps; non-square maps; ragged last splits
                                   (3, 64, 64, 56, 56), (2, 32, 128, 56, 56), (2, 64, 64, 6, 56), (16, 64, 64, 56, 56),
                                   (5, 128, 128, 28, 28), (3, 64, 192, 28, 28), (2, 32, 64, 20, 28),
                                   (5, 256, 256, 14, 14), (4, 32, 64, 14, 14), (3, 64, 64, 6, 14), (64, 256, 256, 14, 14),
                                   (7, 512, 512, 7, 7), (3, 96, 64, 7, 7), (256, 512, 512, 7, 7)], ids=lambda s: 'x'.join(map(str, s)))
def test_winograd_only_shapes(K, shape):
    """Shapes only the Winograd F(3x3, 2x2) instances take: ragged image groups on 4-wide maps (any N: V3's 66 images, the
    stacked branches' 132), Ci a multiple of 32, the ImageNet-geometry map widths 56 / 28 / 14 / 7."""
    if K.conv_algo() != 'winograd':
        pytest.skip('the Winograd instances')
    n, ci, co, h, w = shape
    x, dy = _rand((n, ci, h, w), 31 + n), _rand((n, co, h, w), 32 + co)
    got = K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1)
    assert got is not None
    ref = _ref(x, dy, (co, ci, 3, 3))
    assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(got, K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1))


def test_one_hot_small_integers_are_exact_in_the_winograd_kernel(K):
    """Small-integer operands on every map width: the transforms only add (and halve at the very end), so every dW entry is
    exact -- a wrong tap, a halo that is not zero, a tile pair read from the wrong place or a sign folded the wrong way is an
    exact mismatch.  (Runs in every mode: the direct kernels are exact here too.)"""
    for hw in (4, 8, 16, 32, 56, 28, 14, 7):
        n, ci, co = 5, 32, 64
        rs = np.random.RandomState(hw)
        x = torch.zeros(n, ci, hw, hw, device=DEV)
        dy = torch.zeros(n, co, hw, hw, device=DEV)
        for _ in range(400):
            x[rs.randint(n), rs.randint(ci), rs.choice([0, hw - 1, rs.randint(hw)]), rs.choice([0, hw - 1, rs.randint(hw)])] = float(rs.randint(1, 5))
            dy[rs.randint(n), rs.randint(co), rs.choice([0, hw - 1, rs.randint(hw)]), rs.choice([0, hw - 1, rs.randint(hw)])] = float(rs.randint(1, 4))
        got = K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1)
        if got is None:                                       # (direct instances need Ci a multiple of 64)
            continue
        ref = _ref(x, dy, (co, ci, 3, 3))
        assert float(ref.abs().sum()) > 0 and torch.equal(got.double(), ref), hw


def test_1x1_stride1_one_hot_small_integers_are_exact(K):
    """Small integers at sparse positions, every chunk geometry of the 1x1 stride-1 kernel: each dW entry is an exact
    small sum -- a position read from the wrong plane, a pad that is not zero, a group pair skipped or taken twice, or a
    32 x 32 block written to the wrong slot of the partial tiles is an exact mismatch."""
    for hw, n, ci, co in ((56, 3, 128, 64), (28, 3, 64, 192), (14, 5, 128, 128), (7, 4, 192, 128)):
        rs = np.random.RandomState(hw)
        x = torch.zeros(n, ci, hw, hw, device=DEV)
        dy = torch.zeros(n, co, hw, hw, device=DEV)
        for _ in range(3000):
            i, j = rs.choice([0, hw - 1, rs.randint(hw)]), rs.choice([0, hw - 1, rs.randint(hw)])
            x[rs.randint(n), rs.randint(ci), i, j] = float(rs.randint(1, 5))
            dy[rs.randint(n), rs.randint(co), i, j] = float(rs.randint(1, 4))
        got = K.conv_wgrad(x, dy, (co, ci, 1, 1), 1, 0)
        assert got is not None
        ref = _ref(x, dy, (co, ci, 1, 1), 1, 0)
        assert float(ref.abs().sum()) > 0 and torch.equal(got.double(), ref), hw


def test_rank2_term_fused_into_the_1x1_stride1_reduction_equals_the_separate_update(K):
    n, ci, co, h = 6, 512, 256, 7
    x, dy = _rand((n, ci, h, h), 5), _rand((n, co, h, h), 6)
    dg, db = _rand((co,), 7), _rand((co,), 8)
    m = torch.rand(2, ci, dtype=torch.float64, device=DEV) * 2 - 1
    fused = K.conv_wgrad(x, dy, (co, ci, 1, 1), 1, 0, dg, db, m)
    plain = K.conv_wgrad(x, dy, (co, ci, 1, 1), 1, 0)
    assert torch.equal(fused, K.gamma_beta_bwd_acc(dg, db, m, plain.clone()))


def test_rank2_term_fused_into_the_1x1_reduction_equals_the_separate_update(K):
    n, ci, co, h = 8, 256, 512, 8
    x, dy = _rand((n, ci, h, h), 5), _rand((n, co, h // 2, h // 2), 6)
    dg, db = _rand((co,), 7), _rand((co,), 8)
    m = torch.rand(2, ci, dtype=torch.float64, device=DEV) * 2 - 1
    fused = K.conv_wgrad(x, dy, (co, ci, 1, 1), 2, 0, dg, db, m)
    plain = K.conv_wgrad(x, dy, (co, ci, 1, 1), 2, 0)
    assert torch.equal(fused, K.gamma_beta_bwd_acc(dg, db, m, plain.clone()))


def test_rank2_term_fused_into_the_reduction_equals_the_separate_update(K):
    """The passport branch's weight-gradient term added in the reduction pass is, bit for bit, deepipr_gamma_beta_bwd_acc
    applied to the plain weight gradient."""
    n, c, h = 16, 512, 4
    x, dy = _rand((n, c, h, h), 5), _rand((n, c, h, h), 6)
    dg, db = _rand((c,), 7), _rand((c,), 8)
    m = torch.rand(2, c * 9, dtype=torch.float64, device=DEV) * 2 - 1
    fused = K.conv_wgrad(x, dy, (c, c, 3, 3), 1, 1, dg, db, m)
    plain = K.conv_wgrad(x, dy, (c, c, 3, 3), 1, 1)
    assert torch.equal(fused, K.gamma_beta_bwd_acc(dg, db, m, plain.clone()))


@pytest.mark.parametrize('case', [
    dict(n=4, ci=3, co=64, h=16, w=16), dict(n=4, ci=4, co=64, h=32, w=32), dict(n=4, ci=64, co=64, h=10, w=10), dict(n=4, ci=64, co=64, h=8, w=7), dict(n=4, ci=64, co=96, h=8, w=8),
    dict(n=4, ci=64, co=64, h=5, w=5, k=1, pad=0), dict(n=4, ci=32, co=64, h=8, w=8, k=1, pad=0), dict(n=4, ci=64, co=64, h=64, w=64, stride=2),
    dict(n=4, ci=64, co=64, h=8, w=8, stride=3),
    dict(n=3, ci=64, co=64, h=4, w=4)])
def test_shapes_outside_the_kernel_are_refused_before_anything_is_enqueued(K, case):
    from deepipr_amd import _lib
    if K.conv_algo() == 'winograd' and case == dict(n=3, ci=64, co=64, h=4, w=4):
        pytest.skip('the Winograd instance masks ragged image groups: an odd batch on 4-wide maps is inside the kernel')
    k, stride, pad = case.get('k', 3), case.get('stride', 1), case.get('pad', 1)
    n, ci, co, h, w = case['n'], case['ci'], case['co'], case['h'], case['w']
    assert K.conv_wgrad_workspace(n, ci, co, h, w, k, k, stride, pad) == 0
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    x, dy = _rand((n, ci, h, w), 1), _rand((n, co, oh, ow), 2)
    assert K.conv_wgrad(x, dy, (co, ci, k, k), stride, pad) is None
    dw = torch.empty(co, ci, k, k, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    rc = _lib.lib().deepipr_conv_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, ci, co, h, w, k, k, stride, pad,
                                       None, None, None, ws.data_ptr(), ws.numel(), None)
    assert rc == -3 and b'conv_wgrad' in _lib.lib().deepipr_last_error()


def test_convblock_and_passport_layer_take_the_kernel_and_match_the_library_gradient(K, monkeypatch):
    """Module level: ConvBlock and PassportBlock (data convolution inside the fused node, rank-2 term fused) with the own
    weight gradient against the same modules with DEEPIPR_OWN_WGRAD off (the vendor library's wgrad + the separate
    rank-2 update): same forward bit for bit, weight gradients within 1e-5 of scale."""
    from deepipr_amd import passport_ops as P
    from deepipr_amd.models.layers.conv2d import ConvBlock
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    calls = {'n': 0}
    real = type(K).conv_wgrad

    def counted(self, *a, **k):
        calls['n'] += 1
        return real(self, *a, **k)
    monkeypatch.setattr(type(K), 'conv_wgrad', counted)
    res = {}
    for own in (True, False):
        monkeypatch.setattr(P, 'OWN_WGRAD', own)
        torch.manual_seed(0)
        np.random.seed(0)
        plain = ConvBlock(64, 64, 3, 1, 1).to(DEV)
        pas = PassportBlock(64, 64, 3, 1, 1, {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}).to(DEV)
        x = _rand((16, 64, 16, 16), 3).requires_grad_(True)
        y = pas(plain(x))
        (y.square().mean() + pas.sign_loss.loss).backward()
        res[own] = (y.detach(), plain.conv.weight.grad, pas.weight.grad, x.grad)
    assert calls['n'] == 2                                   # both layers of the `own` pass, none of the other
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1:], res[False][1:]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
