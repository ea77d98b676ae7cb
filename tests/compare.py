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


def states_close(a, b, rtol=1e-3, atol=1e-5, what='state'):
    """Two state_dicts (same keys) of torch tensors: every floating tensor allclose; a failure reports HOW the two
    differ (how many tensors, the largest relative deviations, whether they are bit-identical elsewhere) instead of
    the first key only -- a rounding-level divergence and a read of freed memory look very different here."""
    import torch
    bad, exact = [], 0
    for k in a:
        if not a[k].dtype.is_floating_point:
            continue
        if torch.equal(a[k], b[k]):
            exact += 1
            continue
        if not torch.allclose(a[k], b[k], rtol=rtol, atol=atol):
            d = (a[k].double() - b[k].double()).abs()
            bad.append((float(d.max() / (a[k].double().abs().max() + 1e-30)), k, int((d > atol + rtol * b[k].double().abs()).sum()),
                        a[k].numel(), bool(torch.isfinite(b[k]).all())))
    assert not bad, ('%s: %d tensors differ beyond rtol %g (%d bit-identical); worst (max|d|/max|a|, name, elements off, '
                     'numel, finite): %s' % (what, len(bad), rtol, exact, sorted(bad, reverse=True)[:6]))
