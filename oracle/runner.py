"""Run one golden case on an implementation and collect comparable outputs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The same collector is driven with three
implementations: the real reference (tools/gen_golden.py, build container only), the oracle
(tests/test_oracle_golden.py, CPU) and the HIP build (tests/test_parity_gpu.py, GPU).  An
implementation is an object with

    build(case)                       -> nn.Module on its device, constructed like the reference does
    is_passport(m) / is_private(m)    -> bool
    step(model, optimizer, batch, wm) -> dict of python floats (the trainer's epoch dict)
    device

All of them expose the reference's module API (get_scale / get_bias / b / sign_loss*).
"""
import contextlib
import io
import random

import numpy as np
import torch

from oracle import patterns
from oracle.cases import CASES, SGD
from oracle.np_passport import parse_signature


def _np(t):
    return t.detach().to('cpu', torch.float32).numpy().copy()


def _sig_overrides(model, impl, config_leaves):
    """For layers whose config value is an ASCII string, keep those leading bits after the fill."""
    over = {}
    for name, m in model.named_modules():
        if impl.is_passport(m) and isinstance(config_leaves.get(name), str):
            filled = patterns.tensor_for(name + '.b', m.b.shape)
            over[name + '.b'] = parse_signature(config_leaves[name], m.b.numel(), filled)
    return over


def _leaves(cfg, prefix=''):
    """Config tree -> {module path: leaf}.  AlexNet keys index `features`, ResNet keys are paths."""
    out = {}
    for k, v in cfg.items():
        if isinstance(v, dict):
            out.update(_leaves(v, prefix + k + '.'))
        else:
            out[prefix + k] = v
    return out


def fill(model, impl, case, salt=0):
    leaves = _leaves(case['config'])
    if case['arch'] == 'alexnet':
        leaves = {'features.' + k: v for k, v in leaves.items()}
    sd = model.state_dict()
    keys = set(sd.keys())
    over = _sig_overrides(model, impl, leaves)
    with torch.no_grad():
        for name, t in sd.items():
            if t is None:
                continue
            cname = patterns.canonical(name, keys)
            v = over.get(cname)
            if v is None:
                v = patterns.tensor_for(cname, t.shape, salt)
            t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).to(t.dtype))


def case_inputs(case):
    hw = case.get('hw', 32)
    x, y = patterns.batch(case['n'], 3, hw, hw, case['ncls'])
    wm = None
    if case.get('wm'):
        wm = patterns.batch(2, 3, hw, hw, case['ncls'], salt=1)      # dataset.py:188-191: trigger batch = 2
    return x, y, wm


def collect(case_name, impl, quiet=True):
    case = CASES[case_name]
    dev = impl.device
    out = {}
    torch.manual_seed(0)
    np.random.seed(0)
    model = impl.build(case)
    private = case['scheme'] != 1

    # constructor-time signature parse, before anything is overwritten
    for name, m in model.named_modules():
        if impl.is_passport(m):
            out['ctor_b/' + name] = _np(m.b)

    x, y, wm = case_inputs(case)
    x, y = x.to(dev), y.to(dev)
    if wm is not None:
        wm = (wm[0].to(dev), wm[1].to(dev))

    model.train()
    sink = io.StringIO()
    if case.get('key_type', 'random') == 'random':
        with torch.no_grad():                   # materialise key_type='random' keys
            model(x)
        fill(model, impl, case)
    else:
        # --key-type shuffle (train_v1.py:30-31, experiments/classification.py:68-100): `nkeys` candidate images
        # per key are pushed through a plain net and every passport layer keeps one passport picked from the
        # activations that feed it (passport_generator.set_key -> set_intermediate_keys -> passport_selection,
        # which draws from python's `random`).  Weights first: the keys are buffers the fill must not overwrite.
        fill(model, impl, case)
        plain = patterns.fill_state(impl.plain(case), salt=9)
        hw = case.get('hw', 32)
        kx, _ = patterns.batch(case['nkeys'], 3, hw, hw, case['ncls'], salt=2)
        ky, _ = patterns.batch(case['nkeys'], 3, hw, hw, case['ncls'], salt=3)
        random.seed(4321)
        with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
            impl.set_keys(plain, model, kx.to(dev), ky.to(dev))
        for name, m in model.named_modules():
            if impl.is_passport(m):
                out['key/' + name] = patterns.grad_digest(_np(m.key))
                out['skey/' + name] = patterns.grad_digest(_np(m.skey))

    # per-layer gamma / beta / sign loss before the step
    with torch.no_grad():
        for name, m in model.named_modules():
            if not impl.is_passport(m):
                continue
            if impl.is_private(m):
                g, bta = m.get_scale(ind=1), m.get_bias(ind=1)
                sl = m.sign_loss_private
            else:
                g, bta = m.get_scale(), m.get_bias()
                sl = m.sign_loss
            out['gamma/' + name] = _np(g).reshape(-1)
            out['beta/' + name] = _np(bta).reshape(-1)
            out['bits/' + name] = np.sign(_np(g).reshape(-1)).astype(np.int8)
            if sl is not None:
                out['layer_sign_loss/' + name] = np.float64(float(sl.loss))
                out['layer_sign_acc/' + name] = np.float64(float(sl.acc))

    # one optimisation step through the implementation's own trainer
    logits = []
    hook = model.register_forward_hook(lambda mod, inp, o: logits.append(_np(o)))
    opt = torch.optim.SGD(model.parameters(), **SGD)
    with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
        res = impl.step(model, opt, (x, y), wm)
    hook.remove()
    for k, v in res.items():
        if k != 'time':
            out['train/' + k] = np.float64(v)
    for i, lg in enumerate(logits):
        out['logits_train/%d' % i] = lg

    for name, p in model.named_parameters():
        out['grad/' + name] = patterns.grad_digest(_np(p.grad))
        out['post/' + name] = patterns.grad_digest(_np(p))
    for name, b in model.named_buffers():
        if name.endswith('running_mean') or name.endswith('running_var'):
            out['stat/' + name] = patterns.grad_digest(_np(b))

    # evaluation-mode logits after the step
    model.eval()
    with torch.no_grad():
        if private:
            out['logits_eval/0'] = _np(model(x, ind=0))
            out['logits_eval/1'] = _np(model(x, ind=1))
        else:
            out['logits_eval/0'] = _np(model(x))
        with (contextlib.redirect_stdout(sink) if quiet else contextlib.nullcontext()):
            sig = impl.test_signature(model)
    for k, v in sig.items():
        out['signature/' + k] = np.float64(v)
    return out


def collect_shuttle(shuttle, with_keys=False):
    """Weight-shuttle scenarios (reference experiments/utils.py:100-239), shared by the fixture generator (real
    reference) and the tests (HIP build).  `shuttle` provides
        plain(arch, ncls) / passport(arch, ncls, private) -> nn.Module on shuttle.device,
        n2p(arch, plkeys, passport, plain), n2n(arch, new, old), p2n(arch, plkeys, passport, plain), plkeys(arch).
    Every scenario's result is the digest of every tensor of the destination's state_dict.
    with_keys=True adds the one scenario that needs the device kernels: passport -> plain with gamma/beta derived
    from the keys."""
    out = {}

    def digest(tag, model):
        for k, v in model.state_dict().items():
            out[tag + '/' + k] = patterns.grad_digest(_np(v))

    for arch in ('alexnet', 'resnet18'):
        short = 'alexnet' if arch == 'alexnet' else 'resnet'
        pk = shuttle.plkeys(arch)
        torch.manual_seed(0)
        np.random.seed(0)
        plain = patterns.fill_state(shuttle.plain(arch, 10), salt=3)
        v1 = patterns.fill_state(shuttle.passport(arch, 10, False), salt=4)
        shuttle.n2p(short, pk, v1, plain)
        digest(arch + '_n2p_v1', v1)
        private = patterns.fill_state(shuttle.passport(arch, 10, True), salt=4)
        shuttle.n2p(short, pk, private, plain)
        digest(arch + '_n2p_private', private)
        back = patterns.fill_state(shuttle.plain(arch, 10), salt=5)
        shuttle.p2n(short, pk, v1, back)               # v1 now carries learnable scale/bias: no kernels involved
        digest(arch + '_p2n_learnable', back)
        other = patterns.fill_state(shuttle.plain(arch, 100), salt=6)
        shuttle.n2n(short, other, plain)
        digest(arch + '_n2n', other)
        if with_keys:
            x, _ = patterns.batch(2, 3, 32, 32, 10)
            keyed = shuttle.passport(arch, 10, False)
            keyed.train()
            with torch.no_grad():
                keyed(x.to(shuttle.device))            # materialise key_type='random' keys
            patterns.fill_state(keyed, salt=7)
            for m in keyed.modules():
                if hasattr(m, 'invalidate_key_cache'):
                    m.invalidate_key_cache()
            back = patterns.fill_state(shuttle.plain(arch, 10), salt=5)
            shuttle.p2n(short, pk, keyed, back)
            for k in pk:
                name = ('features.%s' % k) if arch == 'alexnet' else k
                sd = back.state_dict()
                out[arch + '_p2n_keys/' + name + '.bn.weight'] = _np(sd[name + '.bn.weight'])
                out[arch + '_p2n_keys/' + name + '.bn.bias'] = _np(sd[name + '.bn.bias'])
    return out
