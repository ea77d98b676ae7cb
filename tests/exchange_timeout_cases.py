"""The exchange time-out cases of tests/test_norm_kernels_gpu.py::test_exchange_timeout_is_loud, run in a process that
loaded the measurement / test build of the library (DEEPIPR_LIB=.../libdeepipr_hip_trace.so): only that build has the
hooks that force a time-out (deepipr_debug_tune: exchange_drop, exchange_spin)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import patterns                                  # noqa: E402
from oracle.cases import ALPHA, SGD                          # noqa: E402

DEV = 'cuda:0'


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


class raises:
    def __init__(self, match):
        self.match = match

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is RuntimeError and self.match in str(ev), (et, ev)
        return True


def exchange_timeout_poisons_outputs_and_raises(K):
    """A split-channel layer whose partner workgroup never posts its partial sums (test hook) must not carry on with
    stale sums: outputs are NaN, the time-out word is raised, check_exchange() (called by the trainers once per
    epoch) raises.  After re-arming the words the same call is healthy again."""
    from deepipr_amd import _lib
    n, c, h, w = 128, 64, 32, 32
    assert K.allow_sync and K.bn_resident(n, c, h * w) & 1
    rs = np.random.RandomState(0)
    x = dev(rs.standard_normal((n, c, h, w)))
    one, zero = torch.ones(c, device=DEV), torch.zeros(c, device=DEV)

    def run():
        rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        out = K.passport_bn_fwd(x, None, None, one, zero, None, 0.0, True, rm, rv, None, 0.1, 1e-5, True)
        torch.cuda.synchronize()
        return out[0]
    healthy = run()
    assert torch.isfinite(healthy).all() and K.sync_timeouts() == 0
    _lib.debug_tune('exchange_spin', 2000)
    _lib.debug_tune('exchange_drop', 1)
    try:
        y = run()
    finally:
        _lib.debug_tune('exchange_drop', -1)
        _lib.debug_tune('exchange_spin', 0)
    assert torch.isnan(y).all(), 'a timed-out exchange must poison every output of the layer'
    assert K.sync_timeouts() == 1
    with raises('expired in-kernel wait'):
        K.check_exchange()
    assert K.sync_timeouts() == 0                      # re-armed
    again = run()
    assert torch.equal(again, healthy)
    K.check_exchange()


def trainer_stops_on_a_timed_out_exchange(K):
    """Trainer.train raises at the end of the epoch instead of training on with NaN statistics."""
    from deepipr_amd import _lib
    from deepipr_amd.experiments.trainer import Trainer
    from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
    from deepipr_amd.models.resnet_passport import ResNet18Passport
    from oracle.cases import resnet18_config
    kw = construct_passport_kwargs_from_dict({'passport_config': resnet18_config(), 'norm_type': 'bn',
                                              'key_type': 'random', 'sl_ratio': ALPHA})
    torch.manual_seed(0)
    np.random.seed(0)
    net = ResNet18Passport(num_classes=10, passport_kwargs=kw).to(DEV)
    x, y = patterns.batch(128, 3, 32, 32, 10)
    opt = torch.optim.SGD(net.parameters(), **SGD)
    tr = Trainer(net, opt, None, torch.device(DEV))
    _lib.debug_tune('exchange_spin', 2000)
    _lib.debug_tune('exchange_drop', 0)
    try:
        with raises('expired in-kernel wait'):
            tr.train(0, [(x.to(DEV), y.to(DEV))])
    finally:
        _lib.debug_tune('exchange_drop', -1)
        _lib.debug_tune('exchange_spin', 0)
        K.reset_sync_words()
    assert K.sync_timeouts() == 0


if __name__ == '__main__':
    from deepipr_amd import _lib, passport_ops
    assert _lib.has_test_hooks(), 'needs DEEPIPR_LIB = the test build'
    exchange_timeout_poisons_outputs_and_raises(passport_ops.kernels)
    trainer_stops_on_a_timed_out_exchange(passport_ops.kernels)
    print('exchange timeout cases ok')
