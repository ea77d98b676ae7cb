"""What the `-m gpu` test files share: the kernel fixture, host <-> device helpers, MIOpen pinning."""
import numpy as np
import pytest
import torch

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


class pinned_miopen:
    """MIOpen on its deterministic immediate-mode algorithms for the duration of a block."""

    def __enter__(self):
        self.saved = (torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic)
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = False, True

    def __exit__(self, *exc):
        torch.backends.cudnn.benchmark, torch.backends.cudnn.deterministic = self.saved
        return False
