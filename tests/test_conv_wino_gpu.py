"""deepipr_conv_fwd / deepipr_conv_dgrad with the Winograd F(2x2, 3x3) kernels (csrc/deepipr_conv_wino.inc; the default for every
3x3 stride-1 pad-1 convolution) against the oracle: ATen's convolution / convolution_backward evaluated in float64 (what
`self.conv(x)`, models/layers/passportconv2d.py:218 / models/layers/conv2d.py:31, and its autograd backward compute in the
reference).  Bar: 2e-5 of the result's scale (measured ~1e-6: the transforms add a few roundings to the direct sum's),
bit-reproducible, exact on small-integer one-hot operands, and within 1e-5 of scale of the direct implicit GEMM."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def K():
    from deepipr_amd.passport_ops import kernels
    assert torch.cuda.is_available(), 'needs an MI355X'
    before = kernels.set_conv_algo('winograd')
    yield kernels
    kernels.set_conv_algo(before)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _conv64(x, w):
    return torch.ops.aten.convolution(x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)


def _dgrad64(dy, x, w):
    return torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False,
                                               [0, 0], 1, [True, False, False])[0]


# (N, Ci, Co, H, W): every map width, both m-tile heights, the k-group and split-K forms (few tiles), ragged image groups
# (N not a multiple of the 2 / 8 images a workgroup takes on 8- / 4-wide maps), Ci != Co, channel counts that are multiples of
# 8 / 32 only, the config R / config P shard / V3 (66 images) shapes of ResNet18
SHAPES = [
    (128, 64, 64, 32, 32), (32, 64, 64, 32, 32), (2, 64, 128, 32, 32), (1, 32, 32, 4, 32),
    (128, 128, 128, 16, 16), (32, 128, 128, 16, 16), (3, 128, 64, 16, 16), (66, 128, 128, 16, 16),
    (128, 256, 256, 8, 8), (32, 256, 256, 8, 8), (3, 256, 128, 8, 8), (66, 256, 256, 8, 8), (5, 96, 32, 16, 8),
    (128, 512, 512, 4, 4), (32, 512, 512, 4, 4), (66, 512, 512, 4, 4), (4, 192, 64, 4, 4), (13, 160, 32, 8, 4), (8, 512, 512, 4, 4),
    # ImageNet geometry (ResNet18 / ResNet50 at 224 x 224: 56-, 28-, 14- and 7-wide maps; 28 of a workgroup's 32 tile columns
    # busy, odd maps with half-empty last tiles, rows that start on any 4-byte boundary), ragged bands, odd heights
    (4, 64, 64, 56, 56), (3, 128, 128, 28, 28), (5, 256, 256, 14, 14), (6, 512, 512, 7, 7), (3, 64, 32, 7, 7), (2, 64, 96, 10, 56),
    (2, 32, 64, 5, 16), (1, 96, 64, 9, 28), (7, 64, 64, 3, 14), (33, 256, 256, 14, 14), (32, 512, 512, 7, 7),
]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_winograd_forward_and_backward_data_match_the_float64_oracle(K, shape):
    n, ci, co, h, w = shape
    x, wt = _rand((n, ci, h, w), 1 + n), _rand((co, ci, 3, 3), 2 + co, 0.05)
    dy = _rand((n, co, h, w), 3 + ci)
    assert K.conv_is_winograd(n, ci, co, h, w, 3, 1, 1, 0) and K.conv_is_winograd(n, ci, co, h, w, 3, 1, 1, 1)
    y = K.conv_fwd(x, wt, 1, 1)
    assert y is not None and y.shape == (n, co, h, w)
    ref = _conv64(x, wt)
    assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(y, K.conv_fwd(x, wt, 1, 1))
    dx = K.conv_dgrad(dy, wt, x.shape, 1, 1)
    assert dx is not None and dx.shape == x.shape
    ref = _dgrad64(dy, x, wt)
    assert float((dx.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(dx, K.conv_dgrad(dy, wt, x.shape, 1, 1))
    # the pre-transformed form (deepipr_conv_wino_transform_multi + deepipr_conv_fwd_pre / _dgrad_pre: the filters' images
    # G g G^T written once, rows copied global -> LDS): the same arithmetic, so the same bits
    if ci % 32 == 0 and co % 32 == 0:
        assert K.wino_image_bytes(co, ci) == co * ci * 66
        pre = K.wino_transform([wt])[0]
        assert torch.equal(y, K.conv_fwd(x, wt, 1, 1, pre)) and torch.equal(dx, K.conv_dgrad(dy, wt, x.shape, 1, 1, pre))
        fwd_only = K.wino_transform([wt], backward=False)[0]
        assert fwd_only[1] is None and torch.equal(y, K.conv_fwd(x, wt, 1, 1, fwd_only))
    else:
        assert K.wino_image_bytes(co, ci) == 0


@pytest.mark.parametrize('form', ['2,1,1', '1,1,1', '1,2,1', '2,2,1', '1,2,4', '2,1,2', '1,1,8'])
@pytest.mark.parametrize('shape', [(16, 64, 64, 32, 32), (16, 128, 128, 16, 16), (16, 256, 256, 8, 8), (16, 512, 512, 4, 4)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_every_instance_of_the_family(shape, form):
    """The planner's choice (m-tile height, k-groups, K splits) forced through DEEPIPR_WINO_FORM in a fresh process: every
    template instance against float64."""
    import os
    import subprocess
    import sys
    code = ('import torch\nfrom deepipr_amd.passport_ops import kernels as K\n'
            'n, ci, co, h, w = %r\n'
            'g = torch.Generator().manual_seed(5)\n'
            'x = torch.randn(n, ci, h, w, generator=g).cuda(); wt = (torch.randn(co, ci, 3, 3, generator=g) * 0.05).cuda()\n'
            'dy = torch.randn(n, co, h, w, generator=g).cuda()\n'
            'assert K.conv_is_winograd(n, ci, co, h, w, 3, 1, 1, 0)\n'
            'conv = lambda a, b: torch.ops.aten.convolution(a, b, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)\n'
            'y = K.conv_fwd(x, wt, 1, 1); ref = conv(x.double(), wt.double())\n'
            'assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), "fwd"\n'
            'dx = K.conv_dgrad(dy, wt, x.shape, 1, 1)\n'
            'ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), wt.double(), None, [1, 1], [1, 1], [1, 1], False, '
            '[0, 0], 1, [True, False, False])[0]\n'
            'assert float((dx.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), "dgrad"\n'
            'assert torch.equal(y, K.conv_fwd(x, wt, 1, 1)) and torch.equal(dx, K.conv_dgrad(dy, wt, x.shape, 1, 1))\n'
            'pre = K.wino_transform([wt])[0]\n'
            'assert torch.equal(y, K.conv_fwd(x, wt, 1, 1, pre)) and torch.equal(dx, K.conv_dgrad(dy, wt, x.shape, 1, 1, pre)), "pre"\n'
            'print("ok")\n') % (shape,)
    env = dict(os.environ, DEEPIPR_WINO_FORM=form, DEEPIPR_CONV_ALGO='winograd')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


def test_winograd_agrees_with_the_direct_kernel(K):
    """Same operands through both algorithms: within 1e-5 of scale of each other (they differ by the transforms' roundings)."""
    for n, c, hw in ((32, 64, 32), (32, 128, 16), (32, 256, 8), (32, 512, 4)):
        x, wt, dy = _rand((n, c, hw, hw), 11), _rand((c, c, 3, 3), 12, 0.05), _rand((n, c, hw, hw), 13)
        y, dx = K.conv_fwd(x, wt, 1, 1), K.conv_dgrad(dy, wt, x.shape, 1, 1)
        K.set_conv_algo('direct')
        try:
            assert not K.conv_is_winograd(n, c, c, hw, hw, 3, 1, 1, 0)
            yd, dxd = K.conv_fwd(x, wt, 1, 1), K.conv_dgrad(dy, wt, x.shape, 1, 1)
        finally:
            K.set_conv_algo('winograd')
        assert float((y - yd).abs().max()) <= 1e-5 * float(yd.abs().max())
        assert float((dx - dxd).abs().max()) <= 1e-5 * float(dxd.abs().max())


def test_one_hot_operands_are_exact(K):
    """Small-integer one-hot inputs: the transforms only add and halve, so every product and sum is exact -- a wrong tap, a
    halo that is not zero, a tile written to the wrong pixel or a band leaking into its neighbour is an exact mismatch."""
    for hw in (4, 8, 16, 32, 7, 14, 28, 56):
        n, c = (9, 64) if hw <= 32 else (2, 64)
        rs = np.random.RandomState(hw)
        x = torch.zeros(n, c, hw, hw, device=DEV)
        dy = torch.zeros(n, c, hw, hw, device=DEV)
        w = torch.zeros(c, c, 3, 3, device=DEV)
        for _ in range(300):
            x[rs.randint(n), rs.randint(c), rs.choice([0, hw - 1, rs.randint(hw)]), rs.choice([0, hw - 1, rs.randint(hw)])] = float(rs.randint(1, 5))
            dy[rs.randint(n), rs.randint(c), rs.choice([0, hw - 1, rs.randint(hw)]), rs.choice([0, hw - 1, rs.randint(hw)])] = float(rs.randint(1, 5))
        for _ in range(800):
            w[rs.randint(c), rs.randint(c), rs.randint(3), rs.randint(3)] = float(rs.randint(1, 4))
        assert K.conv_is_winograd(n, c, c, hw, hw, 3, 1, 1, 0)
        y, dx = K.conv_fwd(x, w, 1, 1), K.conv_dgrad(dy, w, x.shape, 1, 1)
        ry, rdx = _conv64(x, w), _dgrad64(dy, x, w)
        assert float(ry.abs().sum()) > 0 and float(rdx.abs().sum()) > 0
        assert torch.equal(y.double(), ry) and torch.equal(dx.double(), rdx)


def test_every_tap_and_every_pixel(K):
    """A filter with ONE non-zero tap shifts the image: all nine taps, both directions, on every map width -- the output must
    be the shifted input exactly (zero padded), pixel for pixel."""
    for hw in (4, 8, 16, 32, 7, 14, 28, 56):
        n, c = 2, 32
        x = torch.arange(n * c * hw * hw, device=DEV, dtype=torch.float32).reshape(n, c, hw, hw) % 251.0
        dy = torch.arange(n * c * hw * hw, device=DEV, dtype=torch.float32).reshape(n, c, hw, hw) % 127.0
        for r in range(3):
            for s in range(3):
                w = torch.zeros(c, c, 3, 3, device=DEV)
                w[:, :, r, s] = torch.eye(c, device=DEV) + torch.eye(c, device=DEV).roll(5, 0)
                y = K.conv_fwd(x, w, 1, 1)
                assert torch.equal(y.double(), _conv64(x, w)), (hw, r, s)
                dx = K.conv_dgrad(dy, w, x.shape, 1, 1)
                assert torch.equal(dx.double(), _dgrad64(dy, x, w)), (hw, r, s)


def test_weight_images_of_many_layers_in_one_launch(K):
    """deepipr_conv_wino_transform_multi with more layers than one launch takes (24), mixed sizes: every image equals the one
    a single-layer call writes; the images of a weight keep their addresses from call to call (a replayed hipGraph reads
    them)."""
    ws = [_rand((co, ci, 3, 3), 100 + i, 0.1) for i, (co, ci) in enumerate([(64, 64), (128, 64), (32, 96), (256, 128)] * 7)]
    many = K.wino_transform(ws)
    for w, (uf, ud) in zip(ws, many):
        keep = (uf.clone(), ud.clone(), uf.data_ptr(), ud.data_ptr())
        uf.zero_()
        ud.zero_()
        one = K.wino_transform([w])[0]
        assert one[0].data_ptr() == keep[2] and one[1].data_ptr() == keep[3]
        co, ci = w.shape[:2]
        rows = lambda img: img.view(-1, 132)[:, :128]          # (the 4 floats of pitch per row are never written)
        assert torch.equal(rows(one[0]), rows(keep[0])) and torch.equal(rows(one[1]), rows(keep[1]))
        assert float(rows(one[0]).abs().sum()) > 0


def test_a_train_step_with_and_without_the_weight_images_is_bit_identical(K):
    """ResNet18 V1 step (forward, backward, SGD) with DEEPIPR_WINO_PRE on (wino_weights: one transform launch per step, the
    Winograd forward / backward-data kernels read the images) and off (every workgroup transforms its filters): the same
    losses and the same parameters after two steps, bit for bit."""
    from deepipr_amd import passport_ops as P
    from deepipr_amd.experiments.trainer import train_step_v1
    from tests.gpu_common import pinned_miopen
    from tests.test_parity_gpu import _fullsize_pair
    from oracle.cases import SGD
    runs = []
    before = P.WINO_PRE
    try:
        for pre in (True, False):
            P.WINO_PRE = pre
            prod, _ref, x, y = _fullsize_pair(False, 32, 10)
            x, y = x.to(DEV), y.to(DEV)
            opt = torch.optim.SGD(prod.parameters(), **SGD)
            with pinned_miopen():
                outs = [[float(v) for v in train_step_v1(prod, opt, x, y)] for _ in range(2)]
            torch.cuda.synchronize()
            runs.append((outs, {k: v.clone() for k, v in prod.state_dict().items()}))
    finally:
        P.WINO_PRE = before
    assert runs[0][0] == runs[1][0]
    for k, v in runs[0][1].items():
        assert torch.equal(v, runs[1][1][k]), k
