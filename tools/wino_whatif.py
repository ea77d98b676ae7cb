#!/usr/bin/env python
"""Time the Winograd kernels of one measurement build (DEEPIPR_LIB = a library whose Winograd unit was compiled with
-DWN_WHATIF=<mask>: parts of the main loop left out, results WRONG on purpose -- deepipr_conv_wino.inc).  One JSON line:
microseconds per call (HIP events around back-to-back calls) of forward / backward-data / weight gradient (+ its reduce) for
the four 3x3 stride-1 layers of ResNet18 at batch 128.  Driven by tools/wino_whatif.sh."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepipr_amd.passport_ops import kernels as K      # noqa: E402
from tools.conv_bench import timeit                     # noqa: E402


def main():
    n = int(os.environ.get('WHATIF_BATCH', '128'))
    dev = torch.device('cuda:0')
    rec = {'mask': int(os.environ.get('WHATIF_MASK', '0')), 'N': n}
    shapes = [(64, 56), (128, 28), (256, 14), (512, 7)] if os.environ.get('WHATIF_IMAGENET') == '1' else [(64, 32), (128, 16), (256, 8), (512, 4)]
    for c, hw in shapes:
        x = torch.randn(n, c, hw, hw, device=dev)
        w = torch.randn(c, c, 3, 3, device=dev) * 0.05
        dy = torch.randn(n, c, hw, hw, device=dev)
        rec['fwd_%d' % hw] = round(timeit(lambda: K.conv_fwd(x, w, 1, 1), 30), 1)
        rec['dgr_%d' % hw] = round(timeit(lambda: K.conv_dgrad(dy, w, x.shape, 1, 1), 30), 1)
        rec['wgr_%d' % hw] = round(timeit(lambda: K.conv_wgrad(x, dy, w.shape, 1, 1), 30), 1)
    print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    main()
