"""Deterministic, name-keyed tensor fills (TEST INFRASTRUCTURE, see oracle/__init__.py).

Golden fixtures cannot carry 10-45 MB of weights, so the fixture generator (which runs the real
reference), the oracle and the GPU-side tests all fill a model's state_dict from this module: the
value of every tensor depends only on its state_dict name and shape, through numpy's legacy
Mersenne-Twister RandomState (bit-stable across numpy versions and machines).
"""
import zlib

import numpy as np
import torch


def _rs(name, salt=0):
    return np.random.RandomState((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)


def canonical(name, keys):
    """Aliases the reference registers twice share one fill.

    `<blk>.conv.weight` is the same Parameter as `<blk>.weight` (models/layers/passportconv2d.py:21)
    and `<blk>.sign_loss.b` / `<blk>.sign_loss_private.b` is the same tensor as `<blk>.b` (:41,46)."""
    if name.endswith('.conv.weight') and name[:-len('conv.weight')] + 'weight' in keys:
        return name[:-len('conv.weight')] + 'weight'
    for tail in ('sign_loss.b', 'sign_loss_private.b'):
        if name.endswith(tail):
            return name[:-len(tail)] + 'b'
    return name


def tensor_for(name, shape, salt=0):
    """The deterministic value of state_dict entry `name` (already canonical)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.rsplit('.', 1)[-1]
    rs = _rs(name, salt)
    if leaf in ('running_mean', 'num_batches_tracked'):  # nn defaults, re-asserted so that a
        return np.zeros(shape, dtype=np.float32)         # key-materialising dummy forward leaves
    if leaf == 'running_var':                            # no trace in the norm statistics
        return np.ones(shape, dtype=np.float32)
    if leaf == 'b':
        return np.where(rs.uniform(size=shape) < 0.5, -1.0, 1.0).astype(np.float32)
    if leaf in ('key', 'skey', 'key_private', 'skey_private'):
        return rs.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    if leaf == 'scale':                                  # PassportPrivateBlock public gamma (init 1)
        return (1.0 + 0.25 * rs.standard_normal(shape)).astype(np.float32)
    if len(shape) == 4:                                  # conv weight: kaiming-normal fan_out scale
        fan_out = shape[0] * shape[2] * shape[3]
        return (rs.standard_normal(shape) * np.sqrt(2.0 / fan_out)).astype(np.float32)
    if len(shape) == 2:                                  # linear weight
        return rs.uniform(-1.0, 1.0, size=shape).astype(np.float32) / np.float32(np.sqrt(shape[1]))
    if leaf == 'weight':                                 # affine norm weight
        return (1.0 + 0.25 * rs.standard_normal(shape)).astype(np.float32)
    return (0.1 * rs.standard_normal(shape)).astype(np.float32)   # biases


def fill_state(model, salt=0):
    """Overwrite every parameter/buffer of `model` in place from its state_dict name."""
    sd = model.state_dict()
    keys = set(sd.keys())
    with torch.no_grad():
        for name, t in sd.items():
            if t is None:
                continue
            v = tensor_for(canonical(name, keys), t.shape, salt)
            if v is not None:
                t.copy_(torch.from_numpy(v).to(t.dtype))
    return model


def batch(n, c, h, w, ncls, salt=0):
    """Synthetic image batch x ~ N(0,1) and labels, keyed on the shape."""
    rs = _rs('batch.%d.%d.%d.%d.%d' % (n, c, h, w, ncls), salt)
    x = rs.standard_normal((n, c, h, w)).astype(np.float32)
    y = rs.randint(0, ncls, size=(n,)).astype(np.int64)
    return torch.from_numpy(x), torch.from_numpy(y)


def grad_digest(g):
    """Small, order-sensitive summary of a gradient tensor: sum, abs-sum and 8 strided samples."""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    idx = (np.arange(8) * max(1, g.size // 8) + (g.size // 16)) % g.size
    return np.concatenate([[g.sum(), np.abs(g).sum()], g[idx]])
