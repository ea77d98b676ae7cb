"""Triage: the ImageNet stem's output error on the whole-net test's pattern data (own kernel / vendor library against float64),
relative to the per-channel spread of the output -- what the following BatchNorm divides by."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import patterns
from deepipr_amd.passport_ops import kernels as K

DEV = 'cuda:0'
x, _ = patterns.batch(32, 3, 224, 224, 1000)
x = x.to(DEV)
conv = torch.nn.Conv2d(3, 64, 7, 2, 3, bias=False).to(DEV)
patterns.fill_state(conv)
w = conv.weight.detach()
ref = torch.ops.aten.convolution(x.double(), w.double(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1)
own = K.conv_fwd(x, w, 2, 3)
lib = torch.nn.functional.conv2d(x, w, None, 2, 3)
std = ref.std(dim=(0, 2, 3), keepdim=True)
print('output scale %.3e, per-channel std min %.3e max %.3e' % (float(ref.abs().max()), float(std.min()), float(std.max())))
for name, y in (('own', own), ('library', lib)):
    e = (y.double() - ref).abs()
    print('%-8s max abs err %.3e (%.2e of scale); max err / channel std %.3e; mean err / std %.3e'
          % (name, float(e.max()), float(e.max() / ref.abs().max()), float((e / std).max()), float((e / std).mean())))
    worst = (e / std).flatten().argmax()
    idx = [int(v) for v in torch.unravel_index(worst, e.shape)]
    print('         worst at', idx, 'ref %.6e got %.6e' % (float(ref.flatten()[worst]), float(y.flatten()[worst])))
# sum of |terms| at the worst spot: how much cancellation
absx = torch.ops.aten.convolution(x.double().abs(), w.double().abs(), None, [2, 2], [3, 3], [1, 1], False, [0, 0], 1)
print('sum |terms| max %.3e, typical ratio sum|terms| / |result| %.1f' % (float(absx.max()), float((absx / ref.abs().clamp_min(1e-30)).median())))
