"""Pin the oracle: oracle/torch_ref.py and oracle/np_passport.py must reproduce what the REAL
reference produced (tests/golden/*.npz, written by tools/gen_golden.py in the build container)."""
import numpy as np
import pytest
import torch

from oracle import np_passport as npp
from oracle import runner
from oracle.cases import CASES
from tests.impls import OracleImpl, load_golden

# CPU-vs-CPU, same ATen kernels: only summation-order noise is tolerated.
RTOL, ATOL = 2e-5, 2e-6


def _close(a, b, name, rtol=RTOL, atol=ATOL):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, name
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    assert np.allclose(a, b, rtol=rtol, atol=atol * scale), '%s: max|d|=%g' % (name, np.abs(a - b).max())


@pytest.mark.parametrize('name', list(CASES))
def test_oracle_matches_reference(name, golden_dir):
    torch.set_num_threads(8)
    gold = load_golden(golden_dir, name)
    got = runner.collect(name, OracleImpl())
    ref_keys = {k for k in gold if not k.startswith('train/acc')}
    for k in sorted(ref_keys):
        if k.startswith('ctor_b/'):
            continue
        if k.startswith('train/') and k not in got:
            continue                         # trainer bookkeeping the oracle step does not report
        assert k in got, k
        if k.startswith('bits/'):
            assert np.array_equal(got[k], gold[k]), k          # signature bits: exact
        else:
            _close(got[k], gold[k], k)


def test_ctor_signature_parse(golden_dir):
    """ASCII signature -> leading +-1 bits (passportconv2d.py:28-38)."""
    gold = load_golden(golden_dir, 'alexnet_v1_sig')
    cfg = CASES['alexnet_v1_sig']['config']
    for idx in ('4', '6'):
        text = cfg[idx]
        ref_b = gold['ctor_b/features.' + idx]
        mine = npp.parse_signature(text, ref_b.size, np.ones(ref_b.size))
        n = len(text) * 8
        assert np.array_equal(mine[:n], ref_b[:n])
        assert set(np.unique(ref_b[n:])) <= {-1.0, 1.0}
        assert npp.decode_signature(ref_b)[:len(text)] == text
    with pytest.raises(Exception, match='Too much bit information'):
        npp.parse_signature('x' * 9, 64, np.ones(64))
    assert np.array_equal(npp.parse_signature(-1, 4, np.ones(4)), -np.ones(4))


def test_numpy_layer_matches_reference_block(golden_dir):
    """Layer-level: key batch 3, stride 2, no ReLU -- numpy restatement vs the reference block."""
    gold = load_golden(golden_dir, 'blocks')
    rs = np.random.RandomState(7)
    w = (rs.standard_normal((16, 8, 3, 3)).astype(np.float32) * 0.2)
    key = rs.uniform(-1, 1, (3, 8, 9, 9)).astype(np.float32)
    skey = rs.uniform(-1, 1, (3, 8, 9, 9)).astype(np.float32)
    x = rs.standard_normal((5, 8, 9, 9)).astype(np.float32)
    cot = rs.standard_normal((5, 16, 5, 5)).astype(np.float32)
    b = np.where(rs.uniform(size=16) < 0.5, -1.0, 1.0).astype(np.float32)
    alpha = 0.5
    W, K, SK, X, COT, B = [a.astype(np.float64) for a in (w, key, skey, x, cot, b)]
    gamma, beta = npp.gamma_beta_fwd(W, SK, K, 2, 1)
    xc = npp.conv2d(X, W, 2, 1)
    y = npp.affine_relu_fwd(xc, gamma, beta, relu=False)
    loss, acc, bits = npp.sign_loss_fwd(gamma, B, alpha)
    _close(gamma, gold['bk3/gamma'], 'gamma')
    _close(beta, gold['bk3/beta'], 'beta')
    _close(y, gold['bk3/y'], 'y')
    _close(loss, gold['bk3/sign_loss'], 'loss')
    assert acc == pytest.approx(float(gold['bk3/sign_acc']))
    # backward: dW gets the data conv, gamma and beta contributions (three-way)
    dxc, dgamma, dbeta = npp.affine_relu_bwd(COT, xc, gamma, beta, relu=False)
    dgamma = dgamma + npp.sign_loss_bwd(1.0, gamma, B, alpha)
    dW_pass = npp.gamma_beta_bwd(dgamma, dbeta, W, SK, K, 2, 1)
    col = npp.im2col(X, 3, 3, 2, 1)
    dW_data = np.einsum('bol,bkl->ok', dxc.reshape(5, 16, -1), col).reshape(W.shape)
    _close(dW_pass + dW_data, gold['bk3/dW'], 'dW', rtol=1e-4, atol=1e-5)
    # pooled-patch identity used by the HIP kernel: gamma = W . s / n
    s, n = npp.pooled_patch_sum(SK, 3, 3, 2, 1)
    _close(W.reshape(16, -1) @ s / n, gold['bk3/gamma'], 'pooled gamma')
    # d/dkey against finite differences of the forward
    _, dsk, dk = npp.gamma_beta_bwd(dgamma, dbeta, W, SK, K, 2, 1, need_dkey=True)
    eps = 1e-6
    for (arr, grad, which) in ((SK, dsk, 0), (K, dk, 1)):
        idx = (1, 3, 4, 5)
        pert = arr.copy()
        pert[idx] += eps
        g2, b2 = npp.gamma_beta_fwd(W, pert if which == 0 else SK, pert if which == 1 else K, 2, 1)
        fd = ((g2 - gamma) @ dgamma + (b2 - beta) @ dbeta) / eps
        assert fd == pytest.approx(grad[idx], rel=1e-4, abs=1e-8)


@pytest.mark.parametrize('name', ['bk3_s2', 'bn_s1', 'sc_1x1'])
def test_dkey_matches_reference_autograd(name, golden_dir):
    """d loss / d key, d skey: the reference's own autograd with the keys turned into nn.Parameters
    (passport_attack_3.py:232-243; goldens blocks.npz: dkey/*) against (1) the oracle's stock-ATen layer and
    (2) the numpy oracle's analytic dkey_of, which is what the HIP kernel is compared with on the GPU."""
    from oracle import torch_ref
    from oracle.cases import DKEY_GEOMETRIES, dkey_inputs
    gold = load_golden(golden_dir, 'blocks')
    ci, co, ks, s, pd, bk, hw, n, norm, relu = DKEY_GEOMETRIES[name]
    t = {k: torch.from_numpy(v) for k, v in dkey_inputs(name).items()}
    torch.manual_seed(0)
    blk = torch_ref.PassportLayerRef(ci, co, ks, s, pd, {'norm_type': norm, 'key_type': 'random', 'sign_loss': 0.5},
                                     relu=relu)
    with torch.no_grad():
        blk.weight.copy_(t['w'])
        blk.b.copy_(t['b'])
    del blk.key, blk.skey
    blk.register_parameter('key', torch.nn.Parameter(t['key'].clone()))
    blk.register_parameter('skey', torch.nn.Parameter(t['skey'].clone()))
    blk.train()
    x = t['x'].clone().requires_grad_(True)
    # keep the gradients that reach gamma / beta: the inputs of the analytic d/dkey
    grads = {}
    orig_pooled = blk._pooled

    def pooled(key):
        r = orig_pooled(key)
        r.register_hook(lambda g, which=('skey' if key is blk.skey else 'key'): grads.__setitem__(which, g.clone()))
        return r
    blk._pooled = pooled
    y = blk(x)
    ((y * t['cot']).sum() + blk.sign_loss.loss).backward()
    pre = 'dkey/' + name + '/'
    _close(y.detach().numpy(), gold[pre + 'y'], 'y')
    _close(blk.key.grad.numpy(), gold[pre + 'dkey'], 'dkey (torch oracle)', rtol=1e-4, atol=1e-6)
    _close(blk.skey.grad.numpy(), gold[pre + 'dskey'], 'dskey (torch oracle)', rtol=1e-4, atol=1e-6)
    _close(blk.weight.grad.numpy(), gold[pre + 'dW'], 'dW', rtol=1e-4, atol=1e-5)
    dgamma = grads['skey'].reshape(-1).double().numpy()
    dbeta = grads['key'].reshape(-1).double().numpy()
    _, dsk, dk = npp.gamma_beta_bwd(dgamma, dbeta, t['w'].double().numpy(), t['skey'].double().numpy(),
                                    t['key'].double().numpy(), s, pd, need_dkey=True)
    _close(dk, gold[pre + 'dkey'], 'dkey (numpy oracle)', rtol=1e-4, atol=1e-6)
    _close(dsk, gold[pre + 'dskey'], 'dskey (numpy oracle)', rtol=1e-4, atol=1e-6)


def test_reference_signs_on_near_zero_rows_document_the_noise_floor(golden_dir):
    """The adversarial fixture (goldens blocks.npz: nearzero/*): above 8 eps * sum|W_k m_k| the reference's own fp32
    gamma has the sign of the exact sum on every row; below it the reference flips some signs itself.  This is the
    regime split the GPU test (test_passport_layer_gpu.py) relies on."""
    gold = load_golden(golden_dir, 'blocks')
    w, skey, g_ref = gold['nearzero/w'], gold['nearzero/skey'], gold['nearzero/gamma_ref']
    co = w.shape[0]
    s, n = npp.pooled_patch_sum(skey.astype(np.float64), 3, 3, 1, 1)
    wm = w.reshape(co, -1).astype(np.float64)
    exact = wm @ (s / n)
    g64, _ = npp.gamma_beta_fwd(w.astype(np.float64), skey.astype(np.float64), skey.astype(np.float64), 1, 1)
    _close(g64, exact, 'pooled identity', rtol=1e-12, atol=1e-13)       # f64 noise of two summation orders
    bound = 8 * 6e-8 * (np.abs(wm) * np.abs(s / n)).sum(axis=1)
    meaningful = np.abs(exact) >= bound
    assert np.array_equal(np.sign(g_ref[meaningful]), np.sign(exact[meaningful]))
    assert np.all(np.abs(g_ref - exact) <= bound)                       # the reference stays within its error bound
    assert (np.sign(g_ref[~meaningful]) != np.sign(exact[~meaningful])).sum() >= 1
