"""deepipr_pooled_linear_fwd / _bwd -- the classifier of the CIFAR-geometry nets, logits = Linear(avg_pool(x)) in one launch per
direction -- against the oracle: the reference's three ops (models/resnet_passport.py:127-129 there: F.avg_pool2d(out, 4), view,
self.linear) and their autograd backward evaluated in float64.  Bar: 1e-5 of scale (sums of 16 and of 512 - 2048 fp32 products),
bit-reproducible; the routing (passport_ops.pooled_linear) falls back to the library ops for everything the kernel does not take."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def P():
    from deepipr_amd import passport_ops
    assert torch.cuda.is_available(), 'needs an MI355X'
    return passport_ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(DEV)


def _ref(x, w, b, dl):
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = None if b is None else b.double().requires_grad_(True)
    out = torch.nn.functional.avg_pool2d(x64, x.shape[2:])
    out = torch.nn.functional.linear(out.view(out.size(0), -1), w64, b64)
    out.backward(dl.double())
    return out.detach(), x64.grad, w64.grad, None if b is None else b64.grad


# (N, C, H, W, classes, bias): config R (128 x 512 x 4x4 -> 10), the config-P shard and V3 batches (100 classes), ragged class
# blocks (K not a multiple of 8), a wider map, the Bottleneck width, no bias, a single image
SHAPES = [(128, 512, 4, 4, 10, True), (32, 512, 4, 4, 100, True), (66, 512, 4, 4, 100, True), (5, 64, 2, 2, 3, True),
          (7, 2048, 4, 4, 100, True), (16, 512, 8, 8, 10, True), (128, 512, 4, 4, 10, False), (1, 128, 4, 4, 128, True),
          (3, 512, 2, 8, 17, True)]


@pytest.mark.parametrize('shape', SHAPES, ids=lambda s: 'x'.join(map(str, s)))
def test_head_matches_the_float64_oracle_and_is_bit_reproducible(P, shape):
    n, c, h, w, k, bias = shape
    K = P.kernels
    assert K.pooled_linear_supported(n, c, h * w, k)
    x, wt = _rand((n, c, h, w), 1 + n), _rand((k, c), 2 + k, 0.05)
    b = _rand((k,), 3 + c, 0.1) if bias else None
    dl = _rand((n, k), 4 + n + k, 0.01)
    logits, pooled = K.pooled_linear_fwd(x, wt, b)
    dx, dw, db = K.pooled_linear_bwd(dl, wt, pooled, tuple(x.shape), bias)
    r_logits, r_dx, r_dw, r_db = _ref(x, wt, b, dl)
    for got, ref in ((logits, r_logits), (dx, r_dx), (dw, r_dw)) + (((db, r_db),) if bias else ()):
        assert got.shape == ref.shape
        assert float((got.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert db is not None or not bias
    l2, p2 = K.pooled_linear_fwd(x, wt, b)
    dx2, dw2, db2 = K.pooled_linear_bwd(dl, wt, p2, tuple(x.shape), bias)
    assert torch.equal(logits, l2) and torch.equal(dx, dx2) and torch.equal(dw, dw2) and (not bias or torch.equal(db, db2))


def test_head_is_exact_on_small_integers(P):
    """Integer maps whose 16-pixel sums are multiples of 16 and small integer weights: every mean, product and sum is exact in
    fp32 -- a wrong channel, class or image index is an exact mismatch."""
    rs = np.random.RandomState(3)
    n, c, k = 9, 128, 11
    x = torch.from_numpy(rs.randint(-4, 5, size=(n, c, 1, 1)).astype(np.float32)).to(DEV).expand(n, c, 4, 4).contiguous()
    wt = torch.from_numpy(rs.randint(-3, 4, size=(k, c)).astype(np.float32)).to(DEV)
    b = torch.from_numpy(rs.randint(-3, 4, size=(k,)).astype(np.float32)).to(DEV)
    dl = torch.from_numpy((16 * rs.randint(-2, 3, size=(n, k))).astype(np.float32)).to(DEV)
    logits, pooled = P.kernels.pooled_linear_fwd(x, wt, b)
    dx, dw, db = P.kernels.pooled_linear_bwd(dl, wt, pooled, tuple(x.shape), True)
    r_logits, r_dx, r_dw, r_db = _ref(x, wt, b, dl)
    assert torch.equal(logits.double(), r_logits) and torch.equal(dx.double(), r_dx)
    assert torch.equal(dw.double(), r_dw) and torch.equal(db.double(), r_db)


def test_routing_and_fallbacks(P, monkeypatch):
    """pooled_linear(): the own kernel for a plain Linear on a supported shape (autograd end to end, equal to the library ops
    within fp32 rounding); the library ops for 1000 classes, 7x7 maps, a hooked module, autocast, DEEPIPR_OWN_HEAD=0."""
    lin = torch.nn.Linear(512, 10).to(DEV)
    x = _rand((32, 512, 4, 4), 11).requires_grad_(True)
    calls = {'n': 0}
    inner = P._PooledLinear.apply

    def counting(*a):
        calls['n'] += 1
        return inner(*a)
    monkeypatch.setattr(P._PooledLinear, 'apply', staticmethod(counting))
    out = P.pooled_linear(lin, x)
    assert calls['n'] == 1
    out.square().sum().backward()
    g_own = (x.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    x.grad = None
    lin.zero_grad(set_to_none=True)
    monkeypatch.setattr(P, 'OWN_HEAD', False)
    ref = P.pooled_linear(lin, x)
    assert calls['n'] == 1
    ref.square().sum().backward()
    assert float((out - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    for a, b in zip(g_own, (x.grad, lin.weight.grad, lin.bias.grad)):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
    monkeypatch.setattr(P, 'OWN_HEAD', True)
    big = torch.nn.Linear(512, 1000).to(DEV)
    assert P.pooled_linear(big, x.detach()).shape == (32, 1000) and calls['n'] == 1           # 1000 classes: the library
    x7 = _rand((4, 512, 7, 7), 12)
    assert P.pooled_linear(lin, x7).shape == (4, 10) and calls['n'] == 1                        # 49 positions: the library
    hooked = torch.nn.Linear(512, 10).to(DEV)
    seen = []
    hooked.register_forward_hook(lambda m, i, o: seen.append(1))
    P.pooled_linear(hooked, x.detach())
    assert seen == [1] and calls['n'] == 1                                                      # the hook ran: the module call
    with torch.autocast('cuda', dtype=torch.bfloat16):
        P.pooled_linear(lin, x.detach())
    assert calls['n'] == 1
    P.pooled_linear(lin, x.detach())
    assert calls['n'] == 2
