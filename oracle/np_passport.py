"""numpy restatement of one DeepIPR passport layer (TEST INFRASTRUCTURE, see oracle/__init__.py).

Everything here is plain numpy on the host.  Arithmetic is carried in the dtype
of the inputs (pass float64 arrays for a "truth" run, float32 to mimic the
reference's precision).  Each function cites the reference lines it restates
(paths relative to /root/reference).
"""
import numpy as np


# --------------------------------------------------------------------------- signature bits
def parse_signature(b, o, rand_sign):
    """Signature vector of a passport layer.  models/layers/passportconv2d.py:25-40.

    b          None | int | str : the `b` entry of passport_kwargs
    o          number of output channels
    rand_sign  length-o array of +-1 used where the reference draws
               torch.sign(torch.rand(o) - 0.5) (:25 and :31)
    An int gives b*ones (:26-27).  A str is expanded to 8 bits per character,
    MSB first, '0' -> -1 and '1' -> +1, written over the leading channels; the
    rest stay random (:28-40).  More than o bits raises (:29-30).
    """
    rand_sign = np.asarray(rand_sign, dtype=np.float32)
    if b is None:
        return rand_sign.copy()
    if isinstance(b, int):
        return np.ones(o, dtype=np.float32) * b
    if isinstance(b, str):
        if len(b) * 8 > o:
            raise Exception('Too much bit information')
        out = rand_sign.copy()
        bits = ''.join(format(ord(c), 'b').zfill(8) for c in b)
        for i, c in enumerate(bits):
            out[i] = -1.0 if c == '0' else 1.0
        return out
    return np.asarray(b, dtype=np.float32)


def decode_signature(bits):
    """Inverse of the str branch above: +-1 per channel -> ASCII text (8 bits per char)."""
    bits = np.asarray(bits).reshape(-1)
    n = (bits.size // 8) * 8
    chars = []
    for i in range(0, n, 8):
        v = 0
        for j in range(8):
            v = (v << 1) | (1 if bits[i + j] > 0 else 0)
        chars.append(chr(v))
    return ''.join(chars)


# --------------------------------------------------------------------------- conv as im2col
def conv_out_hw(h, w, kh, kw, stride, pad):
    return (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1


def im2col(x, kh, kw, stride, pad):
    """x[B,Ci,H,W] -> col[B, Ci*kh*kw, Ho*Wo], k index = (ci*kh + r)*kw + s (nn.Conv2d weight order)."""
    b, ci, h, w = x.shape
    ho, wo = conv_out_hw(h, w, kh, kw, stride, pad)
    xp = np.zeros((b, ci, h + 2 * pad, w + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + h, pad:pad + w] = x
    col = np.zeros((b, ci, kh, kw, ho, wo), dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            col[:, :, r, s] = xp[:, :, r:r + stride * ho:stride, s:s + stride * wo:stride]
    return col.reshape(b, ci * kh * kw, ho * wo)


def conv2d(x, w, stride, pad):
    """Bias-free cross-correlation, what nn.Conv2d(i, o, ks, s, pd, bias=False) computes
    (models/layers/passportconv2d.py:18,148,169,218)."""
    co, ci, kh, kw = w.shape
    b, _, h, ww = x.shape
    ho, wo = conv_out_hw(h, ww, kh, kw, stride, pad)
    col = im2col(x, kh, kw, stride, pad)
    out = np.einsum('ok,bkl->bol', w.reshape(co, -1), col)
    return out.reshape(b, co, ho, wo)


def pooled_patch_sum(key, kh, kw, stride, pad):
    """s[k] = sum over (b, l) of im2col(key)[b, k, l] and the count B*L.

    The build's kernel uses gamma[co] = (sum_k W[co,k] * s[k]) / (B*L), which is the
    reference's conv -> mean(hw) -> mean(b) (passportconv2d.py:148-152) with the two
    linear maps commuted."""
    col = im2col(key, kh, kw, stride, pad)
    return col.sum(axis=(0, 2)), col.shape[0] * col.shape[2]


# --------------------------------------------------------------------------- gamma / beta
def pooled_conv(w, key, stride, pad):
    """conv(key, W) -> view(b,c,-1).mean(2) -> mean(0): passportconv2d.py:148-152 (scale) and
    :169-173 (bias).  Returns [Co]."""
    y = conv2d(key, w, stride, pad)
    b, c = y.shape[:2]
    return y.reshape(b, c, -1).mean(axis=2).mean(axis=0)


def gamma_beta_fwd(w, skey, key, stride, pad):
    """gamma from `skey` (get_scale, :142-158), beta from `key` (get_bias, :163-175)."""
    return pooled_conv(w, skey, stride, pad), pooled_conv(w, key, stride, pad)


def gamma_beta_bwd(dgamma, dbeta, w, skey, key, stride, pad, need_dkey=False):
    """Backward of gamma_beta_fwd: dW[co,k] = dgamma[co]*mean_col(skey)[k] + dbeta[co]*mean_col(key)[k];
    optionally the gradients w.r.t. the two passport tensors (passport_attack_3.py:232-243 makes the
    keys nn.Parameters)."""
    co, ci, kh, kw = w.shape
    ss, ns = pooled_patch_sum(skey, kh, kw, stride, pad)
    sb, nb = pooled_patch_sum(key, kh, kw, stride, pad)
    dw = np.outer(dgamma, ss / ns) + np.outer(dbeta, sb / nb)
    dw = dw.reshape(w.shape)
    if not need_dkey:
        return dw

    def dkey_of(dvec, k):
        b, _, h, ww = k.shape
        ho, wo = conv_out_hw(h, ww, kh, kw, stride, pad)
        u = (dvec @ w.reshape(co, -1)).reshape(ci, kh, kw) / (b * ho * wo)
        gp = np.zeros((b, ci, h + 2 * pad, ww + 2 * pad), dtype=w.dtype)
        for r in range(kh):
            for s in range(kw):
                gp[:, :, r:r + stride * ho:stride, s:s + stride * wo:stride] += u[None, :, r, s, None, None]
        return gp[:, :, pad:pad + h, pad:pad + ww]

    return dw, dkey_of(dgamma, skey), dkey_of(dbeta, key)


# --------------------------------------------------------------------------- affine (+ ReLU)
def affine_relu_fwd(xhat, gamma, beta, relu=True):
    """y = gamma*xhat + beta, then ReLU if the block has one.  passportconv2d.py:220-222."""
    y = gamma.reshape(1, -1, 1, 1) * xhat + beta.reshape(1, -1, 1, 1)
    return np.maximum(y, 0) if relu else y


def affine_relu_bwd(dy, xhat, gamma, beta, relu=True):
    """Backward of the line above.  ReLU'(0) = 0 (threshold_backward), mask recomputed from xhat."""
    g = gamma.reshape(1, -1, 1, 1)
    if relu:
        mask = (g * xhat + beta.reshape(1, -1, 1, 1)) > 0
        dz = np.where(mask, dy, 0).astype(dy.dtype)
    else:
        dz = dy
    dxhat = dz * g
    dgamma = (dz * xhat).sum(axis=(0, 2, 3))
    dbeta = dz.sum(axis=(0, 2, 3))
    return dxhat, dgamma, dbeta


# --------------------------------------------------------------------------- sign loss
MARGIN = 0.1   # models/losses/sign_loss.py:27
L2 = 0.00001   # models/losses/sign_loss.py:53


def sign_loss_fwd(gamma, b, alpha):
    """loss = sum(alpha*relu(-b*gamma + 0.1)) + 1e-5*sum(gamma^2); acc = mean(sign(b)==sign(gamma)).
    models/losses/sign_loss.py:27 (hinge), :53 (L2), :20 (acc).  Also returns sign(gamma) as int8,
    the signature read-out of experiments/trainer_private.py:50,57."""
    gamma = gamma.reshape(-1)
    b = b.reshape(-1)
    hinge = alpha * np.maximum(-b * gamma + gamma.dtype.type(MARGIN), 0)
    loss = hinge.sum() + gamma.dtype.type(L2) * (gamma ** 2).sum()
    acc = (np.sign(b) == np.sign(gamma)).astype(np.float32).mean()
    bits = np.sign(gamma).astype(np.int8)
    return loss, acc, bits


def sign_loss_bwd(dloss, gamma, b, alpha):
    """d loss / d gamma = -alpha*b*[0.1 - b*gamma > 0] + 2e-5*gamma, times the upstream scalar."""
    gamma = gamma.reshape(-1)
    b = b.reshape(-1)
    active = (-b * gamma + gamma.dtype.type(MARGIN)) > 0
    return dloss * (np.where(active, -alpha * b, 0) + 2 * gamma.dtype.type(L2) * gamma)


def signature_detection(gamma, b):
    """mean(sign(gamma) == b): experiments/trainer_private.py:50-53,57-60.  sign(0)=0 never matches +-1."""
    return float((np.sign(gamma.reshape(-1)) == b.reshape(-1)).astype(np.float32).mean())
