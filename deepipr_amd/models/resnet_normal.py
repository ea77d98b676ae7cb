"""Plain ResNets (reference models/resnet_normal.py:9-143): the un-passported baselines and the "pretrained" net
whose activations become intermediate passports.  Only library convolutions plus the build's fused norm / ReLU
kernels; module names (convbnrelu_1, layerN.M.*, linear) follow the reference so state_dicts interchange.
"""
import torch.nn as nn
import torch.nn.functional as F

from deepipr_amd import passport_ops as P
from deepipr_amd.passport_ops import with_wino_weights
from deepipr_amd.models.layers.conv2d import ConvBlock

_STAGE_WIDTHS = (64, 128, 256, 512)
_STAGE_STRIDES = (1, 2, 2, 2)


def _projection(in_planes, out_planes, stride, norm_type, relu):
    """Identity when shapes agree, otherwise the 1x1 strided ConvBlock shortcut."""
    if stride == 1 and in_planes == out_planes:
        return nn.Sequential()
    return ConvBlock(in_planes, out_planes, 1, stride, 0, bn=norm_type, relu=relu)


class BasicBlock(nn.Module):
    """3x3 -> 3x3, every conv block with ReLU (the reference keeps relu=True on convbn_2 and the shortcut)."""
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, norm_type='bn'):
        super().__init__()
        self.convbnrelu_1 = ConvBlock(in_planes, planes, 3, stride, 1, bn=norm_type, relu=True)
        self.convbn_2 = ConvBlock(planes, planes, 3, 1, 1, bn=norm_type, relu=True)
        self.shortcut = _projection(in_planes, planes, stride, norm_type, relu=True)

    def forward(self, x):
        main = self.convbn_2(self.convbnrelu_1(x))
        return P.add_relu(main, self.shortcut(x))


class Bottleneck(nn.Module):
    """1x1 -> 3x3 -> 1x1 (x4); the last conv block and the shortcut carry no ReLU."""
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, norm_type='bn'):
        super().__init__()
        wide = self.expansion * planes
        self.convbnrelu_1 = ConvBlock(in_planes, planes, 1, 1, 0, bn=norm_type, relu=True)
        self.convbnrelu_2 = ConvBlock(planes, planes, 3, stride, 1, bn=norm_type, relu=True)
        self.convbn_3 = ConvBlock(planes, wide, 1, 1, 0, bn=norm_type, relu=False)
        self.shortcut = _projection(in_planes, wide, stride, norm_type, relu=False)

    def forward(self, x):
        main = self.convbn_3(self.convbnrelu_2(self.convbnrelu_1(x)))
        return P.add_relu(main, self.shortcut(x))


class ResNet(nn.Module):
    def __init__(self, block, num_blocks, num_classes=10, norm_type='bn', pretrained=False, imagenet=False):
        super().__init__()
        if pretrained and num_classes == 1000:
            raise NotImplementedError('torchvision-pretrained ImageNet weights are not available offline')
        self.num_blocks = num_blocks
        self.norm_type = norm_type
        if imagenet or num_classes == 1000:             # 7x7/2 stem + 3x3/2 max-pool
            self.convbnrelu_1 = nn.Sequential(ConvBlock(3, 64, 7, 2, 3, bn=norm_type, relu=True),
                                              nn.MaxPool2d(3, 2, 1))
        else:                                           # CIFAR stem
            self.convbnrelu_1 = ConvBlock(3, 64, 3, 1, 1, bn=norm_type, relu=True)
        self.in_planes = 64
        for idx, (width, stride, depth) in enumerate(zip(_STAGE_WIDTHS, _STAGE_STRIDES, num_blocks), start=1):
            blocks = []
            for b in range(depth):
                blocks.append(block(self.in_planes, width, stride if b == 0 else 1, norm_type))
                self.in_planes = width * block.expansion
            setattr(self, 'layer%d' % idx, nn.Sequential(*blocks))
        self.linear = nn.Linear(self.in_planes, num_classes)

    @with_wino_weights
    def forward(self, x):
        out = self.convbnrelu_1(x)
        for idx in (1, 2, 3, 4):
            out = getattr(self, 'layer%d' % idx)(out)
        out = F.adaptive_avg_pool2d(out, (1, 1))
        return self.linear(out.flatten(1))


def _family(block, depths):
    def build(**model_kwargs):
        return ResNet(block, list(depths), **model_kwargs)
    return build


ResNet9 = _family(BasicBlock, (1, 1, 1, 1))
ResNet18 = _family(BasicBlock, (2, 2, 2, 2))
ResNet34 = _family(BasicBlock, (3, 4, 6, 3))
ResNet50 = _family(Bottleneck, (3, 4, 6, 3))
ResNet101 = _family(Bottleneck, (3, 4, 23, 3))
ResNet152 = _family(Bottleneck, (3, 8, 36, 3))
