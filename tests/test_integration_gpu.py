"""INTEGRATION.md section B's reference-side ctypes stub, executed on the GPU as a maintainer of the reference would
use it (first run on hardware in round 3: profiles/r03_integration_stub.log).

What it does: extracts the stub's code block from INTEGRATION.md, points it at the in-tree library and drives
`PassportAffine` + `pooled` forward and backward on a stride-2 3x3 block; the same inputs go through the product's
own binding (deepipr_amd.passport_ops.kernels) -- the two bindings call the same kernels, so every output must be
bit-identical -- and through a stock-ATen composition of models/layers/passportconv2d.py:140-172,218-223 +
models/losses/sign_loss.py:27,53 in float64 (1e-5 of scale).
"""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def stub_namespace():
    from deepipr_amd import _lib
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = next(b for b in blocks if b.lstrip().startswith('# models/layers/_deepipr_hip.py'))
    assert '/path/to/libdeepipr_hip.so' in code
    ns = {}
    exec(compile(code.replace('/path/to/libdeepipr_hip.so', _lib.LIB_PATH), 'INTEGRATION.md#B', 'exec'), ns)
    return ns


def test_reference_side_stub_of_integration_md_runs_on_the_gpu():
    """The code block is EXTRACTED from INTEGRATION.md, so the document cannot drift from what is tested."""
    from deepipr_amd import passport_ops
    K = passport_ops.kernels
    ns = stub_namespace()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    n, ci, co, hin, alpha = 8, 16, 32, 16, 0.1
    conv = torch.nn.Conv2d(ci, co, 3, 2, 1, bias=False).to(dev)
    W = (0.1 * torch.randn(co, ci, 3, 3, generator=g)).to(dev)
    key = torch.randn(1, ci, hin, hin, generator=g).to(dev)
    skey = torch.randn(1, ci, hin, hin, generator=g).to(dev)
    b = torch.sign(torch.randn(co, generator=g)).to(dev)
    xhat = torch.randn(n, co, hin // 2, hin // 2, generator=g).to(dev)
    cot = torch.randn(n, co, hin // 2, hin // 2, generator=g).to(dev)

    # (1) the stub
    w1 = W.clone().requires_grad_(True)
    x1 = xhat.clone().requires_grad_(True)
    m1 = ns['pooled'](skey, key, conv)
    y1, gamma1, loss1, acc1, bits1 = ns['PassportAffine'].apply(x1, w1, m1, b, alpha, True)
    ((y1 * cot).sum() + loss1).backward()

    # (2) the product's binding of the same entry points
    m2 = K.pooled_patch_mean(torch.stack([skey, key]).contiguous(), 3, 3, 2, 1)
    y2, gamma2, beta2, loss2, acc2, bits2 = K.passport_fwd(xhat, W, m2, b, alpha, True)
    one = torch.ones((), device=dev)
    dx2, dw2, _dg, _db = K.passport_bwd(cot.contiguous(), xhat, gamma2, beta2, m2, b, alpha, one, None, None,
                                        tuple(W.shape), True)
    for name, a, c in (('m', m1, m2), ('y', y1, y2), ('gamma', gamma1, gamma2), ('loss', loss1, loss2),
                       ('acc', acc1, acc2), ('bits', bits1, bits2), ('dx', x1.grad, dx2), ('dW', w1.grad, dw2)):
        assert torch.equal(a.detach(), c.detach()), 'stub vs product binding: %s' % name

    # (3) stock ATen in float64, written the way the reference's layer computes it
    w3 = W.double().clone().requires_grad_(True)
    x3 = xhat.double().clone().requires_grad_(True)
    F = torch.nn.functional
    gamma3 = F.conv2d(skey.double(), w3, None, 2, 1).view(1, co, -1).mean(dim=2).view(1, co, 1, 1)
    beta3 = F.conv2d(key.double(), w3, None, 2, 1).view(1, co, -1).mean(dim=2).view(1, co, 1, 1)
    y3 = torch.relu(gamma3 * x3 + beta3)
    hinge3 = (alpha * torch.relu(-gamma3.view(-1) * b.double() + 0.1)).sum()       # sign_loss.py:27
    loss3 = hinge3 + 1e-5 * (gamma3 ** 2).sum()                                      # sign_loss.py:53
    ((y3 * cot.double()).sum() + loss3).backward()

    def close(name, a, ref, tol=1e-5):
        scale = max(1.0, float(ref.abs().max()))
        err = float((a.double() - ref).abs().max())
        assert err <= tol * scale, '%s: %g > %g' % (name, err, tol * scale)
    close('y', y1.detach(), y3.detach())
    close('gamma', gamma1.detach(), gamma3.detach().view(-1))
    close('dx', x1.grad, x3.grad)
    close('dW', w1.grad, w3.grad, tol=1e-4)
    assert torch.equal(bits1.cpu().to(torch.float64), torch.sign(gamma3.detach().view(-1)).cpu())
    assert abs(float(loss1.detach()) - float(loss3.detach())) <= 1e-5 * max(1.0, abs(float(loss3.detach())))


def test_section_c_convolution_stub_runs_on_the_gpu():
    """the reference-side `Conv2dMI355X` Function (EXTRACTED from the document) on the three
    kinds of convolution of a ResNet block -- 3x3 stride 1 (own weight gradient only), 3x3 stride 2 and 1x1 stride 2 (all
    directions) -- against F.conv2d in float64: output, dx, dW within 1e-5 of scale."""
    from deepipr_amd import _lib
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = next(b for b in blocks if b.lstrip().startswith('# models/layers/_deepipr_conv.py'))
    ns = {}
    exec(compile(code.replace('/path/to/libdeepipr_hip.so', _lib.LIB_PATH), 'INTEGRATION.md#C', 'exec'), ns)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(11)
    for ci, co, h, k, st in ((64, 64, 16, 3, 1), (64, 128, 32, 3, 2), (64, 128, 32, 1, 2)):
        pd = k // 2
        x = torch.randn(8, ci, h, h, generator=g).to(dev).requires_grad_(True)
        w = (0.05 * torch.randn(co, ci, k, k, generator=g)).to(dev).requires_grad_(True)
        cot = torch.randn(8, co, h // st, h // st, generator=g).to(dev)
        y = ns['Conv2dMI355X'].apply(x, w, st, pd)
        (y * cot).sum().backward()
        x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
        y64 = torch.nn.functional.conv2d(x64, w64, None, st, pd)
        (y64 * cot.double()).sum().backward()
        for name, a, ref in (('y', y.detach(), y64.detach()), ('dx', x.grad, x64.grad), ('dW', w.grad, w64.grad)):
            err = float((a.double() - ref).abs().max()) / max(1e-12, float(ref.abs().max()))
            assert err <= 1e-5, (ci, co, h, k, st, name, err)


def test_pre_transformed_winograd_stub_of_integration_md_runs_on_the_gpu():
    """INTEGRATION.md section D, extracted and executed: the filter images written once, forward and backward-data from them --
    bit-identical to the product's binding (same kernels) and within 2e-5 of scale of float64 ATen, small-batch deep layer
    (split K over workgroups: the workspace path) included."""
    from deepipr_amd import _lib
    from deepipr_amd.passport_ops import kernels as K
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = next(b for b in blocks if b.lstrip().startswith('# models/layers/_deepipr_wino.py'))
    ns = {}
    exec(compile(code.replace('/path/to/libdeepipr_hip.so', _lib.LIB_PATH), 'INTEGRATION.md#D', 'exec'), ns)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(7)
    for n, ci, co, hw in [(16, 64, 64, 32), (8, 512, 512, 4), (5, 96, 64, 14)]:
        x = torch.randn(n, ci, hw, hw, generator=g).to(dev)
        w = (0.05 * torch.randn(co, ci, 3, 3, generator=g)).to(dev)
        dy = torch.randn(n, co, hw, hw, generator=g).to(dev)
        images = ns['winograd_images']([w])[0]
        y = ns['conv3x3_pre'](x, w.shape, images)
        dx = ns['conv3x3_pre'](x, w.shape, images, dy=dy)
        assert torch.equal(y, K.conv_fwd(x, w, 1, 1)) and torch.equal(dx, K.conv_dgrad(dy, w, x.shape, 1, 1))
        ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
        assert float((y.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
        ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                  [True, False, False])[0]
        assert float((dx.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


def test_section_e_imagenet_stub_runs_on_the_gpu():
    """INTEGRATION.md section E, extracted and executed: the reference-side `Conv1x1MI355X` (1x1 convolutions of the Bottleneck,
    stride 1 and 2, all three directions through the C ABI) against F.conv2d in float64 -- 1e-5 of scale -- and
    `MaxPool3x3s2MI355X` against F.max_pool2d -- bit for bit, forward and backward."""
    from deepipr_amd import _lib
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = next(b for b in blocks if b.lstrip().startswith('# models/layers/_deepipr_imagenet.py'))
    ns = {}
    exec(compile(code.replace('/path/to/libdeepipr_hip.so', _lib.LIB_PATH), 'INTEGRATION.md#E', 'exec'), ns)
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(13)
    for n, ci, co, h, st in ((4, 64, 256, 56, 1), (3, 1024, 256, 14, 1), (5, 512, 2048, 7, 1), (4, 256, 512, 56, 2), (3, 1024, 2048, 14, 2)):
        x = torch.randn(n, ci, h, h, generator=g).to(dev).requires_grad_(True)
        w = (0.05 * torch.randn(co, ci, 1, 1, generator=g)).to(dev).requires_grad_(True)
        cot = torch.randn(n, co, h // st, h // st, generator=g).to(dev)
        y = ns['Conv1x1MI355X'].apply(x, w, st)
        (y * cot).sum().backward()
        x64, w64 = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
        y64 = torch.nn.functional.conv2d(x64, w64, None, st, 0)
        (y64 * cot.double()).sum().backward()
        for name, a, ref in (('y', y.detach(), y64.detach()), ('dx', x.grad, x64.grad), ('dW', w.grad, w64.grad)):
            err = float((a.double() - ref).abs().max()) / max(1e-12, float(ref.abs().max()))
            assert err <= 1e-5, (n, ci, co, h, st, name, err)
    for shape in ((2, 64, 112, 112), (3, 5, 7, 9)):
        x = torch.relu(torch.randn(shape, generator=g)).to(dev).requires_grad_(True)      # the stem's input: a ReLU output, many ties
        x2 = x.detach().clone().requires_grad_(True)
        y = ns['MaxPool3x3s2MI355X'].apply(x)
        y2 = torch.nn.functional.max_pool2d(x2, 3, 2, 1)
        cot = torch.randn(tuple(y2.shape), generator=g).to(dev)
        y.backward(cot)
        y2.backward(cot)
        assert torch.equal(y, y2) and torch.equal(x.grad, x2.grad)
