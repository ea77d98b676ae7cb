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
    for pool in (torch.nn.MaxPool2d(2, 1), torch.nn.MaxPool2d(3, 2, 1, ceil_mode=True), torch.nn.MaxPool2d(3, 1, 1)):
        assert torch.equal(P.max_pool(pool, x), pool(x))
    pool = torch.nn.MaxPool2d(3, 2, 1)
    seen = []
    pool.register_forward_hook(lambda m, i, o: seen.append(1))
    assert torch.equal(P.max_pool(pool, x), F.max_pool2d(x, 3, 2, 1)) and seen == [1]


def _both2(x):
    from deepipr_amd import passport_ops as P
    pool = torch.nn.MaxPool2d(2, 2)
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = P.max_pool(pool, a)
    yb = F.max_pool2d(b, 2, 2)
    g = torch.Generator(device='cpu').manual_seed(9)
    dy = torch.randn(yb.shape, generator=g).to(x.device)
    ya.backward(dy)
    yb.backward(dy)
    return ya.detach(), yb.detach(), a.grad, b.grad


@pytest.mark.parametrize('shape', [(64, 64, 32, 32), (3, 192, 16, 16), (5, 256, 8, 8), (1, 1, 2, 4), (2, 3, 6, 12)], ids=lambda s: 'x'.join(map(str, s)))
def test_maxpool2x2_equals_aten_bit_for_bit_and_really_runs_the_kernel(shape, monkeypatch):
    """deepipr_maxpool2x2s2_fwd / _bwd -- the CIFAR AlexNet's nn.MaxPool2d(2, 2) (models/alexnet_passport.py:30-38 of the reference):
    forward values and gradient bit for bit ATen's, ties (three-level maps, ReLU outputs) and NaN / -inf included."""
    from deepipr_amd import passport_ops as P
    calls = {'n': 0}
    inner = P._MaxPool2x2s2.apply

    def counting(*a):
        calls['n'] += 1
        return inner(*a)
    monkeypatch.setattr(P._MaxPool2x2s2, 'apply', staticmethod(counting))
    g = torch.Generator(device='cpu').manual_seed(sum(shape))
    for x in (torch.randn(shape, generator=g), torch.randint(0, 3, shape, generator=g).float(), torch.relu(torch.randn(shape, generator=g))):
        ya, yb, ga, gb = _both2(x.to(DEV))
        assert ya.shape == yb.shape and torch.equal(ya, yb) and torch.equal(ga, gb)
    assert calls['n'] == 3
    x = torch.randn(shape, generator=g)
    x.view(-1)[::7] = float('nan')
    x.view(-1)[1::5] = float('-inf')
    ya, yb, ga, gb = _both2(x.to(DEV))
    assert torch.equal(torch.isnan(ya), torch.isnan(yb)) and torch.equal(torch.nan_to_num(ya, nan=7.0), torch.nan_to_num(yb, nan=7.0))
    assert torch.equal(ga, gb)


def test_maxpool2x2_shapes_outside_the_kernel_take_the_module():
    from deepipr_amd import passport_ops as P
    for shape in [(2, 3, 7, 8), (2, 3, 8, 6), (2, 3, 5, 5)]:               # odd height, width not a multiple of 4
        x = torch.randn(shape, device=DEV)
        assert torch.equal(P.max_pool(torch.nn.MaxPool2d(2, 2), x), F.max_pool2d(x, 2, 2))
    x = torch.randn(2, 3, 8, 8, device=DEV)
    assert torch.equal(P.max_pool(torch.nn.MaxPool2d(2, 2, 1), x), F.max_pool2d(x, 2, 2, 1))      # padded: the module
