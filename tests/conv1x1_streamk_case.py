"""Helper of tests/test_conv_1x1_gpu.py::test_stream_k_tail_*: the own 1x1 GEMM (k_conv1x1_gemm) on a few shapes under the
DEEPIPR_CONV1X1_STREAMK mode of the environment (0: whole tiles only; 2: the stream-K tail whenever the tile count is not a multiple
of the chip's workgroup slots -- the library reads the switch when it first plans such a call, hence a process per mode).

    python tests/conv1x1_streamk_case.py out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (N, Ci, Co, H, W): tile counts below / above one round of slots, 64- and 128-channel tiles, 16- and 4-byte items, a ragged last
# position tile, K of 2 .. 64 chunks
SHAPES = [(8, 64, 128, 14, 14), (64, 1024, 256, 14, 14), (256, 512, 128, 7, 7), (40, 256, 1024, 14, 14), (3, 2048, 512, 7, 7),
          (130, 128, 64, 28, 28), (9, 64, 64, 5, 9)]


def main():
    from deepipr_amd.passport_ops import kernels as K
    dev = torch.device('cuda:0')
    out = {}
    for i, (n, ci, co, h, w) in enumerate(SHAPES):
        g = torch.Generator(device='cpu').manual_seed(100 + i)
        x = torch.randn(n, ci, h, w, generator=g).to(dev)
        wt = (0.05 * torch.randn(co, ci, 1, 1, generator=g)).to(dev)
        dy = torch.randn(n, co, h, w, generator=g).to(dev)
        xi = (torch.randint(0, 4, (n, ci, h, w), generator=g) * (torch.rand(n, ci, h, w, generator=g) < 0.2)).float().to(dev)
        wi = (torch.randint(-3, 4, (co, ci, 1, 1), generator=g) * (torch.rand(co, ci, 1, 1, generator=g) < 0.3)).float().to(dev)
        di = (torch.randint(0, 4, (n, co, h, w), generator=g) * (torch.rand(n, co, h, w, generator=g) < 0.2)).float().to(dev)
        y, dx = K.conv_fwd(x, wt, 1, 0), K.conv_dgrad(dy, wt, tuple(x.shape), 1, 0)
        yi, dxi = K.conv_fwd(xi, wi, 1, 0), K.conv_dgrad(di, wi, tuple(x.shape), 1, 0)
        again = K.conv_fwd(x, wt, 1, 0)
        ref = torch.nn.functional.conv2d(x.double(), wt.double())
        dref = torch.nn.functional.conv_transpose2d(dy.double(), wt.double())
        out['err_y_%d' % i] = float((y.double() - ref).abs().max() / ref.abs().max())
        out['err_dx_%d' % i] = float((dx.double() - dref).abs().max() / dref.abs().max())
        out['exact_%d' % i] = bool(torch.equal(yi.double(), torch.nn.functional.conv2d(xi.double(), wi.double()))
                                   and torch.equal(dxi.double(), torch.nn.functional.conv_transpose2d(di.double(), wi.double())))
        out['repeat_%d' % i] = bool(torch.equal(y, again))
        out['ws_%d' % i] = (K.conv_workspace(n, ci, co, h, w, 1, 1, 0, 0), K.conv_workspace(n, ci, co, h, w, 1, 1, 0, 1))
        out['y_%d' % i] = y[:2, :3].cpu().numpy()
    torch.cuda.synchronize()
    np.savez(sys.argv[1], **{k: np.asarray(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
