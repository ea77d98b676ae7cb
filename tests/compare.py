import numpy as np


def close(a, b, name, rtol, atol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, '%s: shape %s vs %s' % (name, a.shape, b.shape)
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    assert np.allclose(a, b, rtol=rtol, atol=atol * scale), '%s: max|d|=%g (scale %g)' % (
        name, np.abs(a - b).max(), scale)


def compare_case(got, gold, rtol, atol, skip_prefixes=()):
    """Every array of the golden case must be reproduced; signature bits exactly."""
    for k in sorted(gold):
        if any(k.startswith(p) for p in skip_prefixes):
            continue
        assert k in got, 'missing ' + k
        if k.startswith('bits/'):
            assert np.array_equal(got[k], gold[k]), k
        else:
            close(got[k], gold[k], k, rtol, atol)
