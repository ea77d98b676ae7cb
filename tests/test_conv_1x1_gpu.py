"""The 1x1 convolutions of the Bottleneck blocks (BASELINE config 5; models/resnet_normal.py:30-49) as this library runs them:
forward / backward-data as plain GEMMs over NCHW, the weight gradient on deepipr_conv_1x1.inc's kernel, stride 2 (the
projection shortcuts at ImageNet map widths) behind deepipr_subsample2 / in front of deepipr_upsample2_zero -- against ATen's
convolution / convolution_backward in float64 (what `self.conv(x)`, models/layers/conv2d.py:31, and its autograd backward
compute in the reference).  Bar: 1e-5 of the result's scale; the gather / scatter kernels exactly."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize('shape', [(3, 4, 8, 8), (2, 3, 56, 56), (2, 5, 14, 14), (1, 2, 6, 10), (4, 64, 28, 28), (2, 3, 2, 2)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_gather_and_scatter_are_exact_and_adjoint(shape):
    from deepipr_amd.passport_ops import kernels as K
    x = _rand(shape, 1)
    ys = K.subsample2(x)
    assert torch.equal(ys, x[:, :, ::2, ::2])
    dy = _rand(ys.shape, 2)
    dx = K.upsample2_zero(dy, tuple(x.shape))
    want = torch.zeros_like(x)
    want[:, :, ::2, ::2] = dy
    assert torch.equal(dx, want)


# (N, Ci, Co, H, W of the input, stride): the Bottleneck's 1x1 convolutions at every ImageNet map width, both strides
SHAPES = [(3, 64, 256, 56, 56, 1), (2, 256, 64, 56, 56, 1), (3, 128, 512, 28, 28, 1), (5, 1024, 256, 14, 14, 1), (7, 512, 2048, 7, 7, 1),
          (2, 256, 512, 56, 56, 2), (3, 512, 1024, 28, 28, 2), (5, 1024, 2048, 14, 14, 2), (4, 64, 128, 56, 56, 2),
          (2, 64, 128, 10, 6, 2), (2, 96, 32, 7, 7, 1)]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_conv2d_of_a_1x1_convolution_matches_the_float64_oracle_in_all_three_directions(shape):
    from deepipr_amd import passport_ops as P
    n, ci, co, h, w, st = shape
    conv = torch.nn.Conv2d(ci, co, 1, st, 0, bias=False).to(DEV)
    x = _rand((n, ci, h, w), 3 + n).requires_grad_(True)
    assert P._gemm_1x1(x.shape, conv.weight, st, 0, x)
    y = P.conv2d(conv, x)
    assert y.grad_fn is not None and 'Conv2dOwn' in type(y.grad_fn).__name__
    dy = _rand(tuple(y.shape), 5 + co)
    y.backward(dy)
    x64, w64 = x.detach().double().requires_grad_(True), conv.weight.detach().double().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x64, w64, None, st, 0)
    ref.backward(dy.double())
    for got, want, what in ((y.detach(), ref.detach(), 'y'), (x.grad, x64.grad, 'dx'), (conv.weight.grad, w64.grad, 'dW')):
        scale = float(want.abs().max())
        assert float((got.double() - want).abs().max()) <= 1e-5 * scale, what
    if st == 2:                                       # the pixels a stride-2 1x1 convolution never reads get an exact zero
        mask = torch.ones_like(x.grad, dtype=torch.bool)
        mask[:, :, ::2, ::2] = False
        assert float(x.grad[mask].abs().max()) == 0.0


def test_passport_layer_with_a_1x1_stride2_data_convolution_follows_the_library_path(monkeypatch):
    """Module level (layer4.0.shortcut of the ResNet50 passport variant: 1x1, stride 2, 14 -> 7 wide, passport flag on): the
    fused node with the GEMM / gather route and the rank-2 passport term fused into the 1x1 weight gradient's reduction,
    against the same module with the route switched off (the vendor library's convolution + the separate rank-2 update)."""
    from deepipr_amd import passport_ops as P
    from deepipr_amd.models.layers.passportconv2d import PassportBlock
    res = {}
    for on in (True, False):
        monkeypatch.setattr(P, 'GEMM_1X1', on)
        torch.manual_seed(0)
        np.random.seed(0)
        pas = PassportBlock(128, 256, 1, 2, 0, {'norm_type': 'bn', 'key_type': 'random', 'sign_loss': 0.1}, relu=False).to(DEV)
        x = _rand((6, 128, 14, 14), 3).requires_grad_(True)
        y = pas(x)
        # (a random projection: mean(y^2) of a normalised map is constant in x -- its gradient is rounding noise)
        ((y * _rand(tuple(y.shape), 9)).mean() + pas.sign_loss.loss).backward()
        res[on] = (y.detach(), pas.weight.grad, x.grad)
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())


# (N, Ci, Co, H, W): the own NCHW GEMM (k_conv1x1_gemm) -- 128- / 64-channel output tiles either direction, 16-byte and 4-byte
# activation items (7x7 planes, odd maps), position tiles that cross image boundaries and a ragged last tile
GEMM_SHAPES = [(3, 64, 256, 56, 56), (2, 256, 64, 56, 56), (5, 128, 512, 28, 28), (3, 1024, 256, 14, 14), (7, 512, 2048, 7, 7),
               (5, 2048, 512, 7, 7), (1, 64, 64, 7, 7), (3, 64, 128, 5, 9), (2, 192, 64, 6, 10), (33, 64, 64, 2, 2), (256, 512, 128, 7, 7)]


@pytest.mark.parametrize('shape', GEMM_SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_own_1x1_gemm_forward_and_backward_data_match_the_float64_oracle_and_are_bit_reproducible(shape):
    from deepipr_amd.passport_ops import kernels as K
    n, ci, co, h, w = shape
    x, wt, dy = _rand((n, ci, h, w), 1 + n), _rand((co, ci, 1, 1), 2 + co, 0.05), _rand((n, co, h, w), 3 + ci)
    assert K.conv_supported(n, ci, co, h, w, 1, 1, 0, 0) and K.conv_supported(n, ci, co, h, w, 1, 1, 0, 1)
    y = K.conv_fwd(x, wt, 1, 0)
    ref = torch.nn.functional.conv2d(x.double(), wt.double())
    assert y.shape == ref.shape and float((y.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(y, K.conv_fwd(x, wt, 1, 0))
    dx = K.conv_dgrad(dy, wt, tuple(x.shape), 1, 0)
    dref = torch.nn.functional.conv_transpose2d(dy.double(), wt.double())
    assert dx.shape == dref.shape and float((dx.double() - dref).abs().max()) <= 1e-5 * float(dref.abs().max())
    assert torch.equal(dx, K.conv_dgrad(dy, wt, tuple(x.shape), 1, 0))


def test_own_1x1_gemm_is_exact_on_small_integers():
    """Sparse small-integer operands: every output is an exact small sum -- a channel pair taken twice or skipped, a position
    decoded into the wrong image, a block stored to the wrong channel rows is an exact mismatch."""
    from deepipr_amd.passport_ops import kernels as K
    for (n, ci, co, hw) in ((3, 128, 64, 14), (4, 64, 192, 7), (2, 64, 128, 28)):
        g = torch.Generator(device='cpu').manual_seed(hw)
        x = (torch.randint(0, 4, (n, ci, hw, hw), generator=g) * (torch.rand(n, ci, hw, hw, generator=g) < 0.2)).float().to(DEV)
        wt = (torch.randint(-3, 4, (co, ci, 1, 1), generator=g) * (torch.rand(co, ci, 1, 1, generator=g) < 0.3)).float().to(DEV)
        dy = (torch.randint(0, 4, (n, co, hw, hw), generator=g) * (torch.rand(n, co, hw, hw, generator=g) < 0.2)).float().to(DEV)
        assert torch.equal(K.conv_fwd(x, wt, 1, 0).double(), torch.nn.functional.conv2d(x.double(), wt.double()))
        assert torch.equal(K.conv_dgrad(dy, wt, tuple(x.shape), 1, 0).double(),
                           torch.nn.functional.conv_transpose2d(dy.double(), wt.double()))


def test_1x1_shapes_outside_the_gemm_kernel_keep_the_blas_route():
    from deepipr_amd.passport_ops import kernels as K
    assert not K.conv_supported(4, 32, 64, 8, 8, 1, 1, 0, 0) and not K.conv_supported(4, 64, 96, 8, 8, 1, 1, 0, 1)
    assert K.conv_fwd(_rand((4, 32, 8, 8), 1), _rand((64, 32, 1, 1), 2), 1, 0) is None


def test_stream_k_tail_of_the_1x1_gemm_is_correct_exact_on_integers_and_reproducible(tmp_path):
    """k_conv1x1_gemm cuts the tiles of the last, partial round of workgroup slots along K (stream-K: partial tiles in a workspace,
    summed in workgroup order by k_conv1x1_tail_sum).  Forced on for every shape whose tile count is not a multiple of the slots
    (DEEPIPR_CONV1X1_STREAMK=2) and switched off (=0), in a process each: both within 1e-5 of scale of float64 ATen, both exact on
    small integers, both bit-reproducible; the forced runs really took the tail (a non-zero workspace) and -- a different summation
    order -- agree with the plain runs to rounding."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    got = {}
    for mode in ('2', '0'):
        out = str(tmp_path / ('streamk%s.npz' % mode))
        subprocess.run([sys.executable, os.path.join(here, 'conv1x1_streamk_case.py'), out], check=True, timeout=600,
                       env=dict(os.environ, DEEPIPR_CONV1X1_STREAMK=mode))
        got[mode] = np.load(out)
    from tests.conv1x1_streamk_case import SHAPES
    took_tail = 0
    for i in range(len(SHAPES)):
        for mode in ('2', '0'):
            d = got[mode]
            assert float(d['err_y_%d' % i]) <= 1e-5 and float(d['err_dx_%d' % i]) <= 1e-5, (SHAPES[i], mode)
            assert bool(d['exact_%d' % i]) and bool(d['repeat_%d' % i]), (SHAPES[i], mode)
        assert tuple(got['0']['ws_%d' % i]) == (0, 0)
        took_tail += int(got['2']['ws_%d' % i][0] > 0) + int(got['2']['ws_%d' % i][1] > 0)
        a, b = got['2']['y_%d' % i], got['0']['y_%d' % i]
        assert np.abs(a - b).max() <= 2e-6 * np.abs(b).max()
    assert took_tail >= len(SHAPES)                            # (a tile count that happens to be a multiple of the slots has no tail)
