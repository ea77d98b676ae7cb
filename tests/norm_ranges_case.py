"""Helper of tests/test_norm_kernels_gpu.py::test_ranges_in_one_launch_equal_one_launch_per_range: one forward + backward of the
fused norm layer (learnable gamma / beta, batch statistics, ReLU) on a map beyond the register file, results to an .npz.
Run once with DEEPIPR_BN_RANGES=1 (all channel ranges in one launch: k_bn_res_fwd_ranges / _bwd_ranges) and once with 0 (one
launch per range, the round-5 form); the library reads the switch when it first plans such a layer, hence a process each.

    python tests/norm_ranges_case.py N C H W out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from deepipr_amd import _lib
    from deepipr_amd.passport_ops import kernels as K
    n, c, h, w = map(int, sys.argv[1:5])
    dev = torch.device('cuda:0')
    rs = np.random.RandomState(n + c + h)
    to = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(dev)
    x = to((rs.standard_normal((n, c, h, w)) * 1.7 + 0.3).astype(np.float32))
    dy = to(rs.standard_normal((n, c, h, w)).astype(np.float32))
    g_in = to((1 + 0.3 * rs.standard_normal(c)).astype(np.float32))
    b_in = to((0.2 * rs.standard_normal(c)).astype(np.float32))
    rm, rv = to(np.zeros(c, np.float32)), to(np.ones(c, np.float32))
    nbt = to(np.array(3, dtype=np.int64), torch.int64)
    lib = _lib.lib()
    passes = (lib.deepipr_passport_bn_passes(n, c, h * w, 0), lib.deepipr_passport_bn_passes(n, c, h * w, 1))
    out = K.passport_bn_fwd(x, None, None, g_in, b_in, None, 0.1, True, rm, rv, nbt, 0.1, 1e-5, True)
    back = K.passport_bn_bwd(dy, x, out[1], None, None, 0.1, None, None, None, None, True, True)
    torch.cuda.synchronize()
    np.savez(sys.argv[5], y=out[0].cpu().numpy(), table=out[1].cpu().numpy(), rm=rm.cpu().numpy(), rv=rv.cpu().numpy(),
             nbt=nbt.cpu().numpy(), dx=back[0].cpu().numpy(), dgamma=back[2].cpu().numpy(), dbeta=back[3].cpu().numpy(),
             passes=np.array(passes), timeouts=np.array(K.sync_timeouts()))


if __name__ == '__main__':
    main()
