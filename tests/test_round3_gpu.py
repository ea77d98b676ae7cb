"""GPU parity tests added in round 3 (every call goes through the C ABI -> HIP kernels).

  * whole-net backward at 1e-4 of scale: ResNet18 V1 (batch 128) and V2 (batch 32) against the float64 oracle with the
    ReLU kinks gated out in BOTH nets (the model-level gradient bars were 2-3 orders looser than
    the north-star number because one flipped mask perturbs everything upstream)
  * the batched passport GEMV / rank-2 update (several layers per launch) against the numpy oracle, bits exact
"""
import numpy as np
import pytest
import torch

from oracle import torch_ref
from tests.test_round2_gpu import pinned_miopen

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def K():
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from deepipr_amd import _lib, passport_ops
    _lib.lib()
    assert type(passport_ops.kernels).__name__ == 'HipKernels'
    return passport_ops.kernels


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


# ----------------------------------------------------------------------------- batched GEMV pair
GEMV_BATCHES = [
    [(512, 2304), (512, 4608), (512, 256), (512, 4608), (512, 4608)],     # ResNet18 layer4 (config R / P)
    [(384, 1728), (256, 3456), (256, 2304)],                              # AlexNet features 4-6 (config A)
    [(5, 7), (64, 75), (33, 1028), (2, 5124), (1, 4)],                    # K % 4 != 0, K beyond one trip, odd row counts
    [(512, 4608)],                                                        # n = 1 (what the single-layer entry points call)
]


@pytest.mark.parametrize('batch', GEMV_BATCHES)
def test_gamma_beta_multi_matches_the_oracle_and_the_single_layer_calls(K, batch):
    """deepipr_gamma_beta_fwd_multi / _bwd_multi: gamma, beta of every layer of a batch in one launch, and the rank-2
    update of all their dW in one launch (fresh and accumulate form) -- against the float64 contraction W . m that
    oracle/np_passport.py: gamma_beta_fwd reduces to for pooled keys (gamma within 1 ulp-ish of the f64 sum: 2e-7 relative to sum |W m|; signature bits exact) and bit-identical to the per-layer
    calls (the same kernel with n = 1)."""
    rs = np.random.RandomState(len(batch) * 1000 + batch[0][1])
    ws = [dev(rs.standard_normal((co, k)) * 0.05) for co, k in batch]
    ms = [dev(rs.uniform(-1, 1, (2, k)), torch.float64) for co, k in batch]
    got = K.gamma_beta_fwd_multi(ws, ms)
    for (g, b), w, m in zip(got, ws, ms):
        w64, m64 = host(w).astype(np.float64), host(m)
        g64, b64 = w64 @ m64[0], w64 @ m64[1]
        mag = np.abs(w64) @ np.abs(m64[0]) + 1e-30
        assert np.all(np.abs(host(g) - g64) <= 2e-7 * mag + 1e-30)
        assert np.all(np.abs(host(b) - b64) <= 2e-7 * (np.abs(w64) @ np.abs(m64[1])) + 1e-30)
        meaningful = np.abs(g64) > 8 * np.finfo(np.float32).eps * mag
        assert np.array_equal(np.sign(host(g))[meaningful], np.sign(g64)[meaningful])       # signature bits
        g1, b1 = K.gamma_beta_fwd(w, m)
        assert torch.equal(g1, g) and torch.equal(b1, b)
    dgs = [dev(rs.standard_normal(co)) for co, k in batch]
    dbs = [dev(rs.standard_normal(co)) for co, k in batch]
    fresh = K.gamma_beta_bwd_multi(dgs, dbs, ms, [torch.full_like(w, float('nan')) for w in ws], False)
    for dw, dg, db, m, w in zip(fresh, dgs, dbs, ms, ws):
        m32 = host(m).astype(np.float32)
        ref = (host(dg)[:, None].astype(np.float64) * m32[0][None, :] + host(db)[:, None].astype(np.float64) * m32[1][None, :])
        assert np.abs(host(dw) - ref).max() <= 2e-6 * (np.abs(ref).max() + 1e-30)
        assert torch.equal(dw, K.gamma_beta_bwd(dg, db, m, tuple(w.shape)))
    base = [dev(rs.standard_normal(tuple(w.shape))) for w in ws]
    acc = K.gamma_beta_bwd_multi(dgs, dbs, ms, [t.clone() for t in base], True)
    for a, t, f in zip(acc, base, fresh):
        assert torch.equal(a, t + f)                                  # one rounding of the same sum


# ----------------------------------------------------------------------------- whole net, tight
class _RecordingF:
    """torch.nn.functional for oracle/torch_ref.py with relu() recording, per layer, which pre-activations sit within
    `tol` of the ReLU kink."""

    def __init__(self, state, tol):
        self.state, self.tol = state, tol

    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    def relu(self, t, *a, **k):
        name = self.state['current']
        if name is not None and t.dim() == 4:
            self.state['near'].setdefault(name, []).append(t.detach().abs() < self.tol)
        return torch.relu(t)


def _layer_modules(net, types):
    return [(k, m) for k, m in net.named_modules() if isinstance(m, types)]


@pytest.mark.miopen_pinned
@pytest.mark.parametrize('private', [False, True])
def test_whole_net_backward_within_1e4_with_relu_kinks_gated(private, monkeypatch):
    """ResNet18 V1 (batch 128) / V2 (batch 32, both branches, one backward) on the GPU against the float64 oracle
    (stock ATen, oracle/torch_ref.py, same weights / keys / batch): logits, losses and EVERY parameter gradient within
    1e-4 of its scale -- the north-star tolerance applied to a whole-net backward pass.

    Two correct implementations may mask an activation that sits within rounding of a ReLU kink differently, and one
    flipped mask perturbs every gradient upstream of it at the 1e-3 level (tests/triage/debug_hooks.py), which is why
    the older model-level tests carry loose bounds.  Here the kinks are removed from the comparison the way
    test_passport_block_backward_chain_at_config_R_shape does for one layer: a first float64 forward finds, per layer,
    the pre-activations within 1e-4 of zero; both nets then run with those elements of the layer's output gated to
    zero (output * keep), so neither value nor gradient passes through a near-kink element.  Gated outputs are 0 or
    >= 1e-4, so the block tails relu(a + b) have no near-kink elements of their own.  The product runs with the
    separate tail kernels (DEEPIPR_TAIL_FUSION=0) so that every layer output exists to be gated; the fused tails are
    bit-identical to them (test_tail_fusion_is_bit_identical_at_model_level)."""
    from deepipr_amd.models._builders import PASSPORT_TYPES
    from deepipr_amd.models.layers.conv2d import ConvBlock
    from tests.test_parity_gpu import _fullsize_pair
    monkeypatch.setenv('DEEPIPR_TAIL_FUSION', '0')
    tol = 1e-4
    n, ncls = (32, 100) if private else (128, 10)
    prod, ref, x, y = _fullsize_pair(private, n, ncls)
    ref = ref.double().to(DEV)
    xg, yg = x.to(DEV), y.to(DEV)
    x64 = xg.double()
    ce = torch.nn.functional.cross_entropy
    inds = (0, 1) if private else (None,)

    def ref_forward():
        return [ref(x64) if i is None else ref(x64, ind=i) for i in inds]

    # ---- pass 1: where are the kinks (float64, no grad; the norm buffers are restored afterwards)
    state = {'current': None, 'near': {}}
    ref_layers = _layer_modules(ref, (torch_ref.ConvBlockRef, torch_ref.PassportLayerRef))
    saved = {k: v.clone() for k, v in ref.state_dict().items()}
    hooks = []
    for name, m in ref_layers:
        hooks.append(m.register_forward_pre_hook(lambda _m, _i, name=name: state.__setitem__('current', name)))
        hooks.append(m.register_forward_hook(lambda _m, _i, _o: state.__setitem__('current', None)))
    monkeypatch.setattr(torch_ref, 'F', _RecordingF(state, tol))
    with torch.no_grad():
        ref_forward()
    monkeypatch.undo()
    monkeypatch.setenv('DEEPIPR_TAIL_FUSION', '0')
    for h in hooks:
        h.remove()
    ref.load_state_dict(saved)
    near = state['near']
    assert set(near) == {k for k, _ in ref_layers} and all(len(v) == len(inds) for v in near.values())
    gated = sum(int(t.sum()) for v in near.values() for t in v)
    total = sum(t.numel() for v in near.values() for t in v)
    assert 0 < gated < 2e-3 * total, (gated, total)

    # ---- pass 2: both nets with the near-kink outputs gated
    def gate(masks, dtype):
        calls = {'n': 0}

        def hook(_m, _i, out):
            keep = (~masks[calls['n'] % len(masks)]).to(dtype)
            calls['n'] += 1
            if isinstance(out, tuple):                          # a layer whose output is handed out twice (the stem)
                return tuple(o * keep for o in out)
            return out * keep
        return hook
    for name, m in ref_layers:
        m.register_forward_hook(gate(near[name], torch.float64))
    prod_layers = dict(_layer_modules(prod, PASSPORT_TYPES + (ConvBlock,)))
    assert set(prod_layers) == set(near)
    # ConvBlocks: a module hook sees the layer's own output (the tail add happens outside the module call).  Passport
    # layers add the residual INSIDE their module call (_forward), so their own output is gated where it is produced.
    from deepipr_amd.models.layers._passport_base import PassportLayerBase
    gates = {}
    for name, m in prod_layers.items():
        if isinstance(m, ConvBlock):
            m.register_forward_hook(gate(near[name], torch.float32))
        else:
            gates[id(m)] = gate(near[name], torch.float32)
    inner = PassportLayerBase._layer

    def gated_layer(self, x_in, force_passport, ind, residual, conv_out=None):
        out = inner(self, x_in, force_passport, ind, residual, conv_out)
        g = gates.get(id(self))
        return out if g is None else g(self, None, out)
    monkeypatch.setattr(PassportLayerBase, '_layer', gated_layer)

    outs_r = ref_forward()
    loss_r = sum(ce(o, yg) for o in outs_r)
    sign_r = sum(m.loss for m in torch_ref.sign_losses(ref) if isinstance(m.loss, torch.Tensor))
    (loss_r + sign_r).backward()
    with pinned_miopen():
        outs_p = [prod(xg) if i is None else prod(xg, ind=i) for i in inds]
        loss_p = sum(ce(o, yg) for o in outs_p)
        if private:
            sign_p = sum(m.sign_loss_private.loss for m in prod.modules() if hasattr(m, 'sign_loss_private'))
        else:
            sign_p = sum(m.sign_loss.loss for m in prod.modules()
                         if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
        (loss_p + sign_p).backward()
        torch.cuda.synchronize()

    for op, orf in zip(outs_p, outs_r):
        scale = max(1.0, float(orf.abs().max()))
        assert float((op.double() - orf).abs().max()) <= 1e-4 * scale
    assert abs(float(loss_p) - float(loss_r)) <= 1e-4 * max(1.0, abs(float(loss_r)))
    assert abs(float(sign_p) - float(sign_r)) <= 1e-4 * max(1.0, abs(float(sign_r)))
    gp = dict(prod.named_parameters())
    worst = (0.0, None)
    for name, p in ref.named_parameters():
        assert p.grad is not None and gp[name].grad is not None, name
        scale = float(p.grad.abs().max()) + 1e-30
        rel = float((gp[name].grad.double() - p.grad).abs().max()) / scale
        worst = max(worst, (rel, name))
        assert rel <= 1e-4, (name, rel, scale)
    print('whole-net backward, kinks gated (%d of %d activations): worst gradient error %.2e of scale (%s)'
          % (gated, total, worst[0], worst[1]))


# ----------------------------------------------------------------------------- cross-entropy head: labels out of range
@pytest.mark.parametrize('bad', [-100, -1, 10, 1 << 40])
def test_fused_cross_entropy_refuses_out_of_range_labels_loudly(K, bad):
    """A label outside [0, C) -- F.cross_entropy's default ignore_index = -100 included, which this head does not
    implement -- must not index out of bounds and must not give a plausible loss: loss is NaN, the row's gradient is
    NaN, the other rows' gradients are untouched (k_ce_rows / k_ce_bwd read row[t] unvalidated)."""
    from deepipr_amd import passport_ops as P
    rs = np.random.RandomState(3)
    logits = dev(rs.standard_normal((16, 10)) * 2).requires_grad_(True)
    target = torch.from_numpy(rs.randint(0, 10, size=16).astype(np.int64)).to(DEV)
    good_loss, _ = P.cross_entropy_top1(logits, target)
    good_loss.backward()
    good = logits.grad.clone()
    logits.grad = None
    target[5] = bad
    loss, top1 = P.cross_entropy_top1(logits, target)
    loss.backward()
    assert torch.isnan(loss) and torch.isfinite(top1)
    g = logits.grad
    assert torch.isnan(g[5]).all()
    keep = torch.ones(16, dtype=torch.bool, device=DEV)
    keep[5] = False
    assert torch.isfinite(g[keep]).all() is not None          # rows scale with dloss = NaN-free upstream gradient (1.0)
    assert torch.equal(g[keep], good[keep])


# ----------------------------------------------------------------------------- maps too large for one pass
LARGE_MAPS = [
    # (N, C, H, W)            forward            backward
    (128, 8, 112, 112),     # one launch, S = 32   2 passes of 4 channels, S = 64
    (64, 64, 56, 56),       # one launch, S = 4    2 passes of 32 channels, S = 8
    (32, 24, 112, 112),     # one launch, S = 8    2 passes of 16 + 8 channels (ragged last pass), S = 16
    (96, 64, 56, 56),       # 2 passes, S = 8      4 passes, S = 16
]


@pytest.mark.parametrize('shape', LARGE_MAPS)
@pytest.mark.parametrize('mode', ['passport', 'public'])
def test_single_pass_in_channel_range_passes_on_large_maps(K, shape, mode):
    """ImageNet-size maps do not fit the register file at once (VERDICT r02 missing #3: they took the three-launch
    form, 32 B per element and step instead of 20).  They now run as channel-range passes of the single-pass kernels,
    every channel split over up to 64 workgroups (deepipr_passport_bn_passes).  Against the float64 oracle
    (tests/oracle_kernels.py): y, table, running statistics, dx, dgamma, dbeta, dW; and against the three-launch form
    of the same library on the same inputs."""
    from deepipr_amd import _lib
    from tests.oracle_kernels import OracleKernels
    O = OracleKernels()
    n, c, h, w = shape
    kk = 36
    lib = _lib.lib()
    passes = (lib.deepipr_passport_bn_passes(n, c, h * w, 0), lib.deepipr_passport_bn_passes(n, c, h * w, 1))
    assert passes[0] >= 1 and passes[1] >= 2, passes
    assert K.bn_resident(n, c, h * w) == 3 and K.bn_slices(n, c, h * w) >= 8
    rs = np.random.RandomState(n + c)
    x = (rs.standard_normal(shape) * 1.7 + 0.3).astype(np.float32)
    dy = rs.standard_normal(shape).astype(np.float32)
    wt = (rs.standard_normal((c, kk)) * 0.05).astype(np.float32)
    m = rs.uniform(-1, 1, (2, kk))
    b = np.where(rs.uniform(size=c) < 0.5, -1.0, 1.0).astype(np.float32)
    g_in = (1 + 0.3 * rs.standard_normal(c)).astype(np.float32)
    b_in = (0.2 * rs.standard_normal(c)).astype(np.float32)
    public = mode == 'public'
    dl = np.array(0.7, dtype=np.float32)

    def run(kern, to):
        rm, rv = to(np.zeros(c, np.float32)), to(np.ones(c, np.float32))
        nbt = to(np.array(3, dtype=np.int64), torch.int64)
        out = kern.passport_bn_fwd(to(x), None if public else to(wt), None if public else to(m, torch.float64),
                                   to(g_in) if public else None, to(b_in) if public else None,
                                   None if public else to(b), 0.1, True, rm, rv, nbt, 0.1, 1e-5, True)
        back = kern.passport_bn_bwd(to(dy), to(x), out[1], None if public else to(m, torch.float64),
                                    None if public else to(b), 0.1, None if public else to(dl), None, None,
                                    None if public else (c, kk), True, True)
        return out, back, rm, rv, nbt
    cpu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt)
    o_gpu, b_gpu, rm_g, rv_g, nbt_g = run(K, dev)
    torch.cuda.synchronize()
    assert K.sync_timeouts() == 0
    _lib.set_resident(False)
    try:
        o_3l, b_3l, rm_3, rv_3, _ = run(K, dev)
    finally:
        _lib.set_resident(True)
    o_ref, b_ref, rm_r, rv_r, nbt_r = run(O, cpu)
    y_r, y_g = o_ref[0].numpy(), host(o_gpu[0])
    bad = np.abs(y_g - y_r) > 2e-5 * (1 + np.abs(y_r))
    assert bad.mean() < 1e-5, bad.sum()
    assert np.abs(host(o_gpu[1])[:, :4] - o_ref[1].numpy()[:, :4]).max() <= 4e-6
    assert np.abs(host(rm_g) - rm_r.numpy()).max() <= 1e-6 and np.abs(host(rv_g) - rv_r.numpy()).max() <= 2e-6
    assert int(nbt_g) == int(nbt_r) == 4                      # counted once, not once per pass
    dx_r, dx_g = b_ref[0].numpy(), host(b_gpu[0])
    scale = np.abs(dx_r).max() + 1e-12
    bad = np.abs(dx_g - dx_r) > 1e-4 * scale
    assert bad.mean() < 1e-4, (bad.sum(), np.abs(dx_g - dx_r).max(), scale)
    for i in (2, 3):
        ref = b_ref[i].numpy()
        assert np.abs(host(b_gpu[i]) - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6), i
    if not public:
        assert abs(float(o_gpu[4]) - float(o_ref[4])) < 2e-5 * max(1, abs(float(o_ref[4])))
        assert np.array_equal(host(o_gpu[6]), o_ref[6].numpy())
        ref = b_ref[1].numpy()
        assert np.abs(host(b_gpu[1]) - ref).max() <= 2e-4 * (np.abs(ref).max() + 1e-6)
    # the two forms of the library on the same inputs: statistics from f64 sums in both -> outputs within an ulp or two
    assert np.abs(host(o_3l[0]) - y_g).max() <= 4e-6 * (1 + np.abs(y_g).max())
    assert np.abs(host(b_3l[0]) - dx_g).max() <= 2e-5 * scale


# ----------------------------------------------------------------------------- external events of a captured step
def test_external_event_of_a_captured_graph_orders_a_side_stream(K):
    """deepipr_event_record on a capturing stream = an external event-record node: after every launch of the graph, a
    stream made to wait for the event (deepipr_stream_wait_event, issued after the launch call) must see everything the
    graph did BEFORE the node -- this is what lets experiments/staged.py start a gradient bucket's all-reduce from
    outside the graph while the replayed backward is still running.  A wait that bound to an older record, or returned
    early, would let the side stream read the previous replay's values."""
    from deepipr_amd import _lib
    dev_ = torch.device(DEV)
    big = torch.zeros(1 << 26, device=dev_)                    # 256 MB: the increment below takes ~100 us
    tail = torch.zeros(1 << 26, device=dev_)
    cap, side = torch.cuda.Stream(device=dev_), torch.cuda.Stream(device=dev_)
    ev = _lib.ExternalEvent()
    cap.wait_stream(torch.cuda.current_stream(dev_))
    with torch.cuda.stream(cap):
        big.add_(0.0)                                          # warm the kernels outside the capture
        tail.add_(0.0)
    torch.cuda.current_stream(dev_).wait_stream(cap)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        big.add_(1.0)
        ev.record(cap)                                         # <- external event-record node
        for _ in range(4):
            tail.add_(1.0)                                     # work behind the node: the graph is still running
    torch.cuda.synchronize()
    for i in range(1, 31):
        g.replay()
        ev.wait(side)
        with torch.cuda.stream(side):
            snap = big[:: 1 << 12].clone()                     # strided sample across the whole buffer
            done = tail[:: 1 << 16].clone()
        side.synchronize()
        assert bool((snap == float(i)).all()), (i, snap.unique().tolist())
        # (not asserted: `done` usually still holds a value below 4 * i -- the side stream ran ahead of the graph's tail)
        assert float(done.max()) <= 4.0 * i
    torch.cuda.synchronize()
    # the host-side wait (what staged.py uses before it enqueues a bucket's collective): after it, what the graph did
    # before the node is visible to a plain read on ANY stream
    for i in range(31, 41):
        g.replay()
        ev.synchronize()
        assert bool((big[:: 1 << 12] == float(i)).all()), i
    torch.cuda.synchronize()


# ----------------------------------------------------------------------------- rank-2 update of a layer group in one launch
@pytest.mark.parametrize('net_kind', ['resnet18_v1', 'resnet18_v2', 'resnet18_v2_private_pass_first', 'alexnet_v1'])
def test_grouped_rank2_update_is_bit_identical_to_the_per_layer_updates(K, net_kind, monkeypatch):
    """The passport branch's dW of a group of layers in ONE launch (_Rank2Group, the default) against one launch per
    layer (DEEPIPR_NO_RANK2_BATCH=1) and against no batching at all (DEEPIPR_NO_GEMV_BATCH=1): logits and every
    parameter gradient bit-identical with MIOpen pinned; the launch counts say which form ran.  The AlexNet's passport
    layers span two backward stages (features 4 | 5, 6): the groups follow the stages."""
    from deepipr_amd import _lib
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from oracle.cases import alexnet_config, resnet18_config
    private = 'v2' in net_kind
    rev = net_kind.endswith('first')     # the private pass's nodes are then the OLDER ones: the other backward order
    if net_kind.startswith('resnet18'):
        from deepipr_amd.models.resnet_passport import ResNet18Passport
        from deepipr_amd.models.resnet_passport_private import ResNet18Private
        cfg, ctor, n = resnet18_config(), (ResNet18Private if private else ResNet18Passport), 32
    else:
        from deepipr_amd.models.alexnet_passport import AlexNetPassport
        cfg, ctor, n = alexnet_config(), AlexNetPassport, 64
    kw = construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn', 'key_type': 'random',
                                              'sl_ratio': 0.1})
    torch.manual_seed(3)
    np.random.seed(3)
    if net_kind.startswith('resnet18'):
        net = ctor(num_classes=10, passport_kwargs=kw).to(DEV)
    else:
        net = ctor(3, 10, kw).to(DEV)
    net.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (n,), generator=g).to(DEV)
    with torch.no_grad():
        net(x)                                                # keys
    state = {k: v.clone() for k, v in net.state_dict().items()}
    ce = torch.nn.functional.cross_entropy

    def step(env):
        for k in ('DEEPIPR_NO_RANK2_BATCH', 'DEEPIPR_NO_GEMV_BATCH'):
            monkeypatch.delenv(k, raising=False)
        for k in env:
            monkeypatch.setenv(k, '1')
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        _lib.profile_enable(True)
        if private:
            outs = [net(x, ind=1), net(x, ind=0)] if rev else [net(x, ind=0), net(x, ind=1)]
            loss = ce(outs[0], y) + ce(outs[1], y) + sum(m.sign_loss_private.loss for m in net.modules()
                                                         if hasattr(m, 'sign_loss_private'))
        else:
            outs = [net(x)]
            loss = ce(outs[0], y) + sum(m.sign_loss.loss for m in net.modules()
                                        if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
        loss.backward()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        prof = _lib.profile_read()
        got = {'logits%d' % i: o.detach().clone() for i, o in enumerate(outs)}
        got.update({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        return got, int(prof['gamma_beta_fwd'][1]), int(prof['gamma_beta_bwd'][1])

    with pinned_miopen():
        grouped, f0, b0 = step(())
        per_layer, f1, b1 = step(('DEEPIPR_NO_RANK2_BATCH',))
        unbatched, f2, b2 = step(('DEEPIPR_NO_GEMV_BATCH',))
    n_layers = 3 if net_kind == 'alexnet_v1' else 5
    fwd_passes = 1                                              # V2: only the private pass (ind = 1) uses the passports
    # layers whose weight gradient deepipr_conv_wgrad computes get their rank-2 term in ITS reduction pass (no launch of
    # their own, grouped or not); the others -- ResNet18: the 1x1 shortcut of layer4.0 -- keep the separate update
    from deepipr_amd import passport_ops as P
    from deepipr_amd.models._builders import PASSPORT_TYPES
    shape = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp: shape.__setitem__(mod, inp[0].shape))
             for m in net.modules() if isinstance(m, PASSPORT_TYPES)]
    with torch.no_grad():
        net(x, ind=1) if private else net(x)
    for h in hooks:
        h.remove()
    net.load_state_dict(state)
    separate = [m for m, shp in shape.items()
                if not P._own_wgrad(torch.empty(shp, device=DEV), m.weight, m.conv.stride[0], m.conv.padding[0])]
    assert len(shape) == n_layers and len(separate) == (0 if net_kind == 'alexnet_v1' else 1)
    assert (f1, b1) == (fwd_passes, len(separate) * fwd_passes), (f1, b1)
    assert (f2, b2) == (n_layers * fwd_passes, len(separate) * fwd_passes), (f2, b2)
    # one launch per group that still has such a layer
    assert (f0, b0) == (fwd_passes, 1 if separate else 0), (f0, b0)
    for name, other in (('per layer', per_layer), ('unbatched', unbatched)):
        assert set(other) == set(grouped)
        diff = {k: float((grouped[k] - other[k]).abs().max()) for k in grouped if not torch.equal(grouped[k], other[k])}
        assert not diff, '%s: %d tensors differ: %s' % (name, len(diff), sorted(diff.items(), key=lambda kv: -kv[1])[:4])


# ----------------------------------------------------------------------------- shared trunk of the V2 / V3 dual forward
@pytest.mark.parametrize('graph', [False, True])
def test_shared_trunk_equals_two_full_passes_on_the_gpu(K, graph, monkeypatch):
    """ResNet18 V2 (config P shard: batch 32, 100 classes): the layers in front of layer4 run once for both branches
    (models/_builders.shared_trunk) against the two full passes (DEEPIPR_NO_SHARED_TRUNK=1), eagerly and replayed from
    the whole-step hipGraph (one eager step / seven replayed ones); logits-derived scalars, every parameter and every buffer (running
    statistics after TWO updates per step, num_batches_tracked) agree -- gradients differ only by the association of
    the two branches' sum."""
    from deepipr_amd.experiments.graph_step import GraphedTrainStep
    from deepipr_amd.experiments.trainer_private import DualBranch, train_step_v23
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.flat_sgd import FlatSGD
    from deepipr_amd.models.resnet_passport_private import ResNet18Private
    from oracle.cases import resnet18_config
    kw = construct_passport_kwargs_from_dict({'passport_config': resnet18_config(), 'norm_type': 'bn',
                                              'key_type': 'random', 'sl_ratio': 0.1})
    g = torch.Generator().manual_seed(2)
    x = torch.randn(32, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 100, (32,), generator=g).to(DEV)
    res = {}
    with pinned_miopen():
        for mode in ('shared', 'twice'):
            if mode == 'twice':
                monkeypatch.setenv('DEEPIPR_NO_SHARED_TRUNK', '1')
            else:
                monkeypatch.delenv('DEEPIPR_NO_SHARED_TRUNK', raising=False)
            torch.manual_seed(4)
            np.random.seed(4)
            net = ResNet18Private(num_classes=100, passport_kwargs=kw).to(DEV)
            net.train()
            with torch.no_grad():
                net(x)
            dual = DualBranch(net)
            opt = FlatSGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
            calls = {'n': 0}
            net.convbnrelu_1.conv.register_forward_hook(lambda *_a: calls.__setitem__('n', calls['n'] + 1))
            if graph:
                net.convbnrelu_1.conv._forward_hooks.clear()       # no Python side effects inside a capture
                step = GraphedTrainStep(train_step_v23, dual, opt, x, y)
                outs = [tuple(float(v) for v in step(x, y)) for _ in range(3)]
            else:
                outs = [tuple(float(v) for v in train_step_v23(dual, opt, x, y))]       # ONE step: see the bars below
            torch.cuda.synchronize()
            res[mode] = dict(outs=outs, state={k: v.detach().clone() for k, v in net.state_dict().items()},
                             calls=calls['n'])
    a, b = res['shared'], res['twice']
    if not graph:
        assert (a['calls'], b['calls']) == (1, 2), (a['calls'], b['calls'])     # the trunk really runs once per step
    if not graph:
        assert a['outs'][0] == b['outs'][0], (a['outs'][0], b['outs'][0])  # first step: identical forward, bit for bit
    for u, v in zip(a['outs'], b['outs']):
        if graph:                                  # top-1 (steps of 100 / 32 %) may flip on a near-tie after seven steps
            u, v = u[:2], v[:2]
        assert np.allclose(u, v, rtol=1e-3 if graph else 1e-4, atol=1e-4 if graph else 1e-5), (u, v)
    for k, v in b['state'].items():
        if k.endswith('num_batches_tracked'):
            assert int(a['state'][k]) == int(v), k
        else:
            # The association of the two branches' sum (a few 1e-6 of the gradient scale) is the only difference after
            # ONE step (eager variant: 1e-8 absolute in the parameters).  From there it grows by orders of magnitude per
            # step -- early training, gradients of O(10) from a sign loss of ~30: 7e-9 after one step, 8e-6 after three,
            # 4e-5 after seven on the CPU; 1e-4 after three on the GPU -- so only the one-step comparison is tight.  The
            # replayed variant has run seven steps by now (three warm-up steps, the captured one, three replays): its bar
            # is the trajectory tests' one.
            worst = float((a['state'][k] - v).abs().max())
            rel, ab = (5e-3, 2e-4) if graph else (1e-4, 5e-5)
            assert worst <= rel * float(v.abs().max()) + ab, (k, worst)


# ----------------------------------------------------------------------------- dual form: two norm layers + tail, one launch
DUAL_SHAPES = [(128, 128, 16, 16), (128, 256, 8, 8), (128, 512, 4, 4), (32, 128, 16, 16), (64, 64, 16, 16),
               (50, 128, 16, 16), (32, 256, 8, 8)]


@pytest.mark.parametrize('relus', [(True, True), (False, False), (True, False)])
@pytest.mark.parametrize('shape', DUAL_SHAPES)
def test_dual_tail_kernels_equal_the_two_separate_fused_layers(K, shape, relus):
    """deepipr_bn_dual_tail_fwd / _bwd (a projection block's last two norm layers + tail, one launch per direction)
    against deepipr_passport_bn_fwd / _bwd called for the shortcut layer and then, with residual / tail_out, for the
    other: out, channel tables, running statistics, dxa, dxb, dgamma, dbeta -- bit for bit; and against float64."""
    n, c, h, w = shape
    if not K.bn_dual_supported(n, c, h * w):
        pytest.skip('shape outside the dual form')
    g = torch.Generator().manual_seed(n + c)
    xa = (torch.randn(n, c, h, w, generator=g) * 1.3 + 0.2).to(DEV)
    xb = (torch.randn(n, c, h, w, generator=g) * 0.7 - 0.1).to(DEV)
    ga, ba, gb, bb = [(torch.randn(c, generator=g) * s + o).to(DEV) for s, o in ((0.5, 1.0), (0.3, 0.0), (0.5, 1.0), (0.3, 0.1))]
    dy = torch.randn(n, c, h, w, generator=g).to(DEV)
    dy2 = torch.randn(n, c, h, w, generator=g).to(DEV)

    def stats():
        return [torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)]
    ra, rb = stats(), stats()
    out, ta, tb = K.bn_dual_tail_fwd(xa, xb, ga, ba, gb, bb, (*ra, 0.1, 1e-5), (*rb, 0.1, 1e-5), *relus)
    dxa, dxb, dga, dba, dgb, dbb = K.bn_dual_tail_bwd(dy, dy2, out, xa, xb, ta, tb, *relus)
    # the separate path: shortcut layer (b) first, then layer a with the tail folded in
    sa, sb = stats(), stats()
    yb, tb2 = K.passport_bn_fwd(xb, None, None, gb, bb, None, 0.0, relus[1], *sb, 0.1, 1e-5, True)[:2]
    out2, ta2 = K.passport_bn_fwd(xa, None, None, ga, ba, None, 0.0, relus[0], *sa, 0.1, 1e-5, True, residual=yb)[:2]
    dxa2, _dw, dga2, dba2, dres = K.passport_bn_bwd(dy, xa, ta2, None, None, 0.0, None, None, None, None, relus[0], True,
                                                    dy2=dy2, tail_out=out2)
    dxb2, _dw, dgb2, dbb2 = K.passport_bn_bwd(dres, xb, tb2, None, None, 0.0, None, None, None, None, relus[1], True)
    torch.cuda.synchronize()
    for name, u, v in (('out', out, out2), ('table_a', ta[:, :4], ta2[:, :4]), ('table_b', tb[:, :4], tb2[:, :4]),
                       ('rm_a', ra[0], sa[0]), ('rv_a', ra[1], sa[1]), ('rm_b', rb[0], sb[0]), ('rv_b', rb[1], sb[1]),
                       ('nbt_a', ra[2], sa[2]), ('nbt_b', rb[2], sb[2]),
                       ('dxa', dxa, dxa2), ('dxb', dxb, dxb2), ('dga', dga, dga2), ('dba', dba, dba2),
                       ('dgb', dgb, dgb2), ('dbb', dbb, dbb2)):
        assert torch.equal(u, v), (name, float((u.double() - v.double()).abs().max()))
    # float64 reference of the whole expression
    A, B = xa.double().requires_grad_(True), xb.double().requires_grad_(True)
    P = [t.double().requires_grad_(True) for t in (ga, ba, gb, bb)]

    def layer(x, gm, bt, relu):
        mu = x.mean((0, 2, 3), keepdim=True)
        var = x.var((0, 2, 3), unbiased=False, keepdim=True)
        y = (x - mu) / torch.sqrt(var + 1e-5) * gm.view(1, -1, 1, 1) + bt.view(1, -1, 1, 1)
        return torch.relu(y) if relu else y
    ref = torch.relu(layer(A, P[0], P[1], relus[0]) + layer(B, P[2], P[3], relus[1]))
    ref.backward((dy + dy2).double())
    assert float((out.double() - ref).abs().max()) < 1e-4
    for name, got, want in (('dxa', dxa, A.grad), ('dxb', dxb, B.grad), ('dga', dga, P[0].grad), ('dba', dba, P[1].grad),
                            ('dgb', dgb, P[2].grad), ('dbb', dbb, P[3].grad)):
        scale = float(want.abs().max()) + 1e-12
        assert float((got.double() - want).abs().max()) <= 2e-4 * scale, (name, scale)


@pytest.mark.parametrize('net_kind', ['resnet18_v1', 'resnet18_v2'])
def test_dual_tail_is_bit_identical_at_model_level(K, net_kind, monkeypatch):
    """ResNet18: layer2.0 and layer3.0 (plain ConvBlocks: convbn_2 + projection shortcut) take the dual form by
    default; DEEPIPR_NO_DUAL_TAIL=1 runs the two layers one after the other.  Logits and every parameter gradient
    bit-identical with MIOpen pinned, two fused launches less per direction."""
    from deepipr_amd import _lib
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.resnet_passport import ResNet18Passport
    from deepipr_amd.models.resnet_passport_private import ResNet18Private
    from oracle.cases import resnet18_config
    private = net_kind.endswith('v2')
    kw = construct_passport_kwargs_from_dict({'passport_config': resnet18_config(), 'norm_type': 'bn',
                                              'key_type': 'random', 'sl_ratio': 0.1})
    torch.manual_seed(3)
    np.random.seed(3)
    net = (ResNet18Private if private else ResNet18Passport)(num_classes=10, passport_kwargs=kw).to(DEV)
    net.train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(128, 3, 32, 32, generator=g).to(DEV)
    y = torch.randint(0, 10, (128,), generator=g).to(DEV)
    with torch.no_grad():
        net(x)
    state = {k: v.clone() for k, v in net.state_dict().items()}
    ce = torch.nn.functional.cross_entropy

    def step(off):
        if off:
            monkeypatch.setenv('DEEPIPR_NO_DUAL_TAIL', '1')
        else:
            monkeypatch.delenv('DEEPIPR_NO_DUAL_TAIL', raising=False)
        net.load_state_dict(state)
        net.zero_grad(set_to_none=True)
        _lib.profile_enable(True)
        if private:
            outs = list(net.forward_dual(x))
            loss = ce(outs[0], y) + ce(outs[1], y) + sum(m.sign_loss_private.loss for m in net.modules()
                                                         if hasattr(m, 'sign_loss_private'))
        else:
            outs = [net(x)]
            loss = ce(outs[0], y) + sum(m.sign_loss.loss for m in net.modules()
                                        if getattr(m, 'sign_loss', None) is not None and hasattr(m, 'conv'))
        loss.backward()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        prof = _lib.profile_read()
        got = {'logits%d' % i: o.detach().clone() for i, o in enumerate(outs)}
        got.update({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None})
        got.update({'buf/' + k: v.clone() for k, v in net.state_dict().items() if 'running' in k})
        return got, int(prof['bn_res_fwd'][1]), int(prof['bn_res_bwd'][1])

    with pinned_miopen():
        dual, f0, b0 = step(False)
        separate, f1, b1 = step(True)
    assert (f1 - f0, b1 - b0) == (2, 2), ((f0, b0), (f1, b1))
    diff = {k: float((dual[k] - separate[k]).abs().max()) for k in dual if not torch.equal(dual[k], separate[k])}
    assert not diff, '%d tensors differ: %s' % (len(diff), sorted(diff.items(), key=lambda kv: -kv[1])[:4])
