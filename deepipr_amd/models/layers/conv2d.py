"""ConvBlock: conv -> norm(affine) -> ReLU for the un-passported layers (reference
models/layers/conv2d.py:5-36).  The convolution stays on the vendor library (MIOpen).  BatchNorm2d(affine) +
ReLU is arithmetically the passport layer's public branch -- relu(weight * xhat + bias) with learnable
weight / bias -- so on the GPU it runs through the same fused kernels (deepipr_passport_bn_fwd/_bwd, W-less
form): one register-resident launch per direction at 8 / 12 B per element (three streaming launches when the
activation does not fit) instead of MIOpen batch-norm + clamp + threshold_backward + batch-norm backward at
~48 B per element.  GroupNorm(affine) / InstanceNorm2d + ReLU likewise through deepipr_passport_gn_*.
`fuse_norm = False` restores the library ops."""
import os

import torch
import torch.nn as nn


# Activations smaller than this stay on the library norm + ReLU kernels (override: DEEPIPR_CONVBLOCK_FUSE_MIN).
# Measured under the default launch mode (hipGraph replay; tools/gpu_fusemin.sh, profiles/r03_fusemin.log, two
# alternating repetitions on one MI355X): 2^18 against round 2's 2^20 -- config P shard 3.081 -> 3.016 ms (-2.1 %: all
# of layer3's ConvBlocks at 32 images per GPU are 2^19 elements), ResNet18 V1 batch 32 -0.5 %, AlexNet V2 batch 64
# -0.8 %, AlexNet V1 -0.3 %; fusing everything (0) measures the same as 2^18, so the cut only keeps toy shapes away.
FUSE_MIN_ELEMENTS = int(os.environ.get('DEEPIPR_CONVBLOCK_FUSE_MIN', 1 << 18))


def make_norm(norm_type, channels, affine):
    """'bn' | 'gn' (channels // 16 groups) | 'in' | anything else -> no norm.
    models/layers/conv2d.py:15-22 and models/layers/passportconv2d.py:56-64."""
    if norm_type == 'bn':
        return nn.BatchNorm2d(channels, affine=affine)
    if norm_type == 'gn':
        return nn.GroupNorm(channels // 16, channels, affine=affine)
    if norm_type == 'in':
        return nn.InstanceNorm2d(channels)          # the reference never makes this one affine
    return None


class ConvBlock(nn.Module):
    def __init__(self, i, o, ks=3, s=1, pd=1, bn='bn', relu=True):
        super().__init__()
        self.conv = nn.Conv2d(i, o, ks, s, pd, bias=(bn == 'none'))
        self.bn = make_norm(bn, o, affine=True)
        self.relu = nn.ReLU(inplace=True) if relu else None
        self.fuse_norm = os.environ.get('DEEPIPR_NO_CONVBLOCK_FUSION') != '1'
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out', nonlinearity='relu')

    def forward_tail(self, x, residual):
        """-> two handles of relu(self(x) + residual): this block as the last one of a residual block; the add and
        the outer ReLU are folded into the norm kernels when they take the single-pass form."""
        from deepipr_amd import passport_ops as P
        y = self(x, residual)                     # through __call__: module hooks see the layer (output: the pair)
        return y if isinstance(y, tuple) else P.add_relu_fork(y, residual)

    def forward_fork(self, x):
        """-> two handles of self(x) for a layer whose output feeds two consumers (the CIFAR stem feeds layer1's first
        conv and its identity shortcut): with the fused norm kernels their gradients are summed inside the backward
        kernel."""
        y = self(x, None, True)
        return y if isinstance(y, tuple) else (y, y)

    def forward(self, x, _residual=None, _fork=False):
        from deepipr_amd import passport_ops as P
        x = P.conv2d(self.conv, x)
        if (self.fuse_norm and x.is_cuda and x.numel() >= FUSE_MIN_ELEMENTS and isinstance(self.bn, nn.BatchNorm2d)
                and self.bn.affine and self.bn.momentum is not None and x.dtype == torch.float32):
            from deepipr_amd import passport_ops as P
            tail = _residual if (_residual is not None and P.bn_tail_fusable(self.bn, x)) else None
            fork = bool(_fork) and tail is None and _residual is None
            return P.bn_affine_relu(x, self.bn.weight, self.bn.bias, self.bn, self.relu is not None, tail, fork)
        if self.fuse_norm and isinstance(self.bn, (nn.GroupNorm, nn.InstanceNorm2d)) and x.is_cuda:
            from deepipr_amd import passport_ops as P
            if P.gn_is_fusable(self.bn, x):          # GroupNorm(affine) / InstanceNorm2d + ReLU in one kernel
                return P.gn_affine_relu(x, getattr(self.bn, 'weight', None), getattr(self.bn, 'bias', None), self.bn,
                                        self.relu is not None)
        if self.bn is not None:
            x = self.bn(x)
        if self.relu is not None:
            x = self.relu(x)
        return x


def _hooked(*modules):
    return any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in modules)


def dual_tail(main, short, x, skip):
    """relu(main(x) + short(skip)) for the last layer and the projection shortcut of a residual block when both are
    plain ConvBlocks (models/resnet_passport.py:67-85): the two convolutions, then ONE fused launch per direction for
    both norm layers and the tail (passport_ops.bn_dual_tail).  -> the pair of output handles, or None when the
    pair does not qualify (the caller then runs the two layers one after the other)."""
    if not (isinstance(main, ConvBlock) and isinstance(short, ConvBlock) and x.is_cuda and x.dtype == torch.float32
            and main.fuse_norm and short.fuse_norm):
        return None
    if _hooked(main, short, main.conv, short.conv, main.bn, short.bn):
        return None                                   # module hooks want to see each layer
    from deepipr_amd import passport_ops as P
    shape = P.conv_out_shape(x, main.conv)
    if shape != P.conv_out_shape(skip, short.conv) or shape[0] * shape[1] * shape[2] * shape[3] < FUSE_MIN_ELEMENTS:
        return None
    if not P.bn_dual_tail_usable(main.bn, short.bn, shape):
        return None
    return P.bn_dual_tail(P.conv2d(main.conv, x), P.conv2d(short.conv, skip), main.bn, short.bn, main.relu is not None,
                          short.relu is not None)
