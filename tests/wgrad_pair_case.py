"""Helper of tests/test_conv_wgrad_gpu.py::test_two_k_ranges_per_workgroup_*: the Winograd weight gradient on a few shapes under the
DEEPIPR_WGRAD_PAIR mode of the environment (0: one K range per workgroup, k_conv_wino_wgrad; 2: two, k_conv_wino_wgrad2, whatever
the planner's rule says -- the library reads the switch when it first plans such a call, hence a process per mode).

    python tests/wgrad_pair_case.py out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (N, Ci, Co, H, W): every width of the kernel, long and short K ranges, an odd count of ranges (the last pair's second half
# idle), ragged image groups on 4-wide maps, Ci of 32
SHAPES = [(128, 64, 64, 32, 32), (5, 64, 64, 32, 32), (128, 128, 128, 16, 16), (7, 64, 128, 16, 16), (128, 256, 256, 8, 8),
          (3, 256, 128, 8, 8), (128, 512, 512, 4, 4), (66, 32, 64, 4, 4), (1, 64, 64, 4, 4)]


def main():
    from deepipr_amd.passport_ops import kernels as K
    dev = torch.device('cuda:0')
    out = {}
    for i, (n, ci, co, h, w) in enumerate(SHAPES):
        g = torch.Generator(device='cpu').manual_seed(300 + i)
        x = torch.randn(n, ci, h, w, generator=g).to(dev)
        dy = torch.randn(n, co, h, w, generator=g).to(dev)
        xi = (torch.randint(0, 4, (n, ci, h, w), generator=g) * (torch.rand(n, ci, h, w, generator=g) < 0.1)).float().to(dev)
        di = (torch.randint(0, 4, (n, co, h, w), generator=g) * (torch.rand(n, co, h, w, generator=g) < 0.1)).float().to(dev)

        def ref(a, b):
            z = torch.zeros(co, ci, 3, 3, dtype=torch.float64, device=dev)
            return torch.ops.aten.convolution_backward(b.double(), a.double(), z, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]
        got, again = K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1), K.conv_wgrad(x, dy, (co, ci, 3, 3), 1, 1)
        r = ref(x, dy)
        out['err_%d' % i] = float((got.double() - r).abs().max() / r.abs().max())
        out['repeat_%d' % i] = bool(torch.equal(got, again))
        out['exact_%d' % i] = bool(torch.equal(K.conv_wgrad(xi, di, (co, ci, 3, 3), 1, 1).double(), ref(xi, di)))
        out['ws_%d' % i] = K.conv_wgrad_workspace(n, ci, co, h, w, 3, 3, 1, 1)
        out['dw_%d' % i] = got[:4, :4].cpu().numpy()
    torch.cuda.synchronize()
    np.savez(sys.argv[1], **{k: np.asarray(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
