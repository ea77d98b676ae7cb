"""deepipr_maxpool3x3s2_fwd / _bwd -- the ImageNet stem's nn.MaxPool2d(3, 2, 1) (models/resnet_passport.py:94-98) -- against
ATen's own max-pool: forward values and the gradient BIT FOR BIT (comparisons only forward; backward adds an input pixel's
up to four window gradients in ATen's order), ties and NaN / -inf included."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _both(x):
    from deepipr_amd import passport_ops as P
    pool = torch.nn.MaxPool2d(3, 2, 1)
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = P.max_pool(pool, a)
    yb = F.max_pool2d(b, 3, 2, 1)
    g = torch.Generator(device='cpu').manual_seed(7)
    dy = torch.randn(yb.shape, generator=g).to(x.device)
    ya.backward(dy)
    yb.backward(dy)
    return ya.detach(), yb.detach(), a.grad, b.grad


@pytest.mark.parametrize('shape', [(2, 3, 112, 112), (3, 5, 7, 9), (1, 1, 1, 1), (2, 4, 8, 8), (1, 2, 2, 3), (4, 64, 56, 56), (2, 3, 15, 16)],
                         ids=lambda s: 'x'.join(map(str, s)))
def test_maxpool_equals_aten_forward_and_backward_bit_for_bit(shape):
    g = torch.Generator(device='cpu').manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(DEV)
    ya, yb, ga, gb = _both(x)
    assert ya.shape == yb.shape and torch.equal(ya, yb)
    assert torch.equal(ga, gb)


def test_maxpool_ties_take_the_first_maximum_like_aten():
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randint(0, 3, (3, 4, 20, 22), generator=g).float().to(DEV)      # three levels: nearly every window has ties
    ya, yb, ga, gb = _both(x)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)
    x = torch.relu(torch.randn(2, 8, 30, 30, generator=g)).to(DEV)            # the stem's real input: a ReLU output, runs of zeros
    ya, yb, ga, gb = _both(x)
    assert torch.equal(ya, yb) and torch.equal(ga, gb)


def test_maxpool_nan_and_minus_infinity_follow_aten():
    g = torch.Generator(device='cpu').manual_seed(5)
    x = torch.randn(2, 3, 12, 12, generator=g)
    x[0, 0, 3, 3] = float('nan')
    x[0, 0, 4, 4] = float('nan')
    x[0, 1] = float('-inf')
    x[1, 2, :, 5] = float('-inf')
    ya, yb, ga, gb = _both(x.to(DEV))
    assert torch.equal(torch.isnan(ya), torch.isnan(yb))
    assert torch.equal(torch.nan_to_num(ya, nan=7.0), torch.nan_to_num(yb, nan=7.0))
    assert torch.equal(ga, gb)


def test_other_pools_and_hooked_pools_stay_with_the_module():
    from deepipr_amd import passport_ops as P
    x = torch.randn(2, 3, 16, 16, device=DEV)
    for pool in (torch.nn.MaxPool2d(2, 2), torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True), torch.nn.MaxPool2d(3, 1, 1)):
        assert torch.equal(P.max_pool(pool, x), pool(x))
    pool = torch.nn.MaxPool2d(3, 2, 1)
    seen = []
    pool.register_forward_hook(lambda m, i, o: seen.append(1))
    assert torch.equal(P.max_pool(pool, x), F.max_pool2d(x, 3, 2, 1)) and seen == [1]
