"""Plain (un-passported) ResNets: the baseline nets and the "pretrained" net whose activations become
intermediate passports (reference models/resnet_normal.py:9-143).  Library ops only."""
import torch.nn as nn
import torch.nn.functional as F

from deepipr_amd import passport_ops as P
from deepipr_amd.models.layers.conv2d import ConvBlock


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, stride=1, norm_type='bn'):
        super().__init__()
        self.convbnrelu_1 = ConvBlock(in_planes, planes, 3, stride, 1, bn=norm_type, relu=True)
        self.convbn_2 = ConvBlock(planes, planes, 3, 1, 1, bn=norm_type, relu=True)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = ConvBlock(in_planes, self.expansion * planes, 1, stride, 0, bn=norm_type, relu=True)

    def forward(self, x):
        return P.add_relu(self.convbn_2(self.convbnrelu_1(x)), self.shortcut(x))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1, norm_type='bn'):
        super().__init__()
        self.convbnrelu_1 = ConvBlock(in_planes, planes, 1, 1, 0, bn=norm_type, relu=True)
        self.convbnrelu_2 = ConvBlock(planes, planes, 3, stride, 1, bn=norm_type, relu=True)
        self.convbn_3 = ConvBlock(planes, self.expansion * planes, 1, 1, 0, bn=norm_type, relu=False)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = ConvBlock(in_planes, self.expansion * planes, 1, stride, 0, bn=norm_type, relu=False)

    def forward(self, x):
        out = self.convbn_3(self.convbnrelu_2(self.convbnrelu_1(x)))
        return P.add_relu(out, self.shortcut(x))


class ResNet(nn.Module):
    def __init__(self, block, num_blocks, num_classes=10, norm_type='bn', pretrained=False, imagenet=False):
        super().__init__()
        if pretrained and num_classes == 1000:
            raise NotImplementedError('torchvision-pretrained ImageNet weights are not available offline')
        self.in_planes = 64
        self.num_blocks = num_blocks
        self.norm_type = norm_type
        if num_classes == 1000 or imagenet:
            self.convbnrelu_1 = nn.Sequential(ConvBlock(3, 64, 7, 2, 3, bn=norm_type, relu=True),
                                              nn.MaxPool2d(3, 2, 1))
        else:
            self.convbnrelu_1 = ConvBlock(3, 64, 3, 1, 1, bn=norm_type, relu=True)
        self.layer1 = self._make_layer(block, 64, num_blocks[0], 1)
        self.layer2 = self._make_layer(block, 128, num_blocks[1], 2)
        self.layer3 = self._make_layer(block, 256, num_blocks[2], 2)
        self.layer4 = self._make_layer(block, 512, num_blocks[3], 2)
        self.linear = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, num_blocks, stride):
        layers = []
        for s in [stride] + [1] * (num_blocks - 1):
            layers.append(block(self.in_planes, planes, s, self.norm_type))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def forward(self, x):
        out = self.layer4(self.layer3(self.layer2(self.layer1(self.convbnrelu_1(x)))))
        out = F.adaptive_avg_pool2d(out, (1, 1))
        return self.linear(out.view(out.size(0), -1))


def _factory(block, depths):
    def make(**model_kwargs):
        return ResNet(block, depths, **model_kwargs)
    return make


ResNet9 = _factory(BasicBlock, [1, 1, 1, 1])
ResNet18 = _factory(BasicBlock, [2, 2, 2, 2])
ResNet34 = _factory(BasicBlock, [3, 4, 6, 3])
ResNet50 = _factory(Bottleneck, [3, 4, 6, 3])
ResNet101 = _factory(Bottleneck, [3, 4, 23, 3])
ResNet152 = _factory(Bottleneck, [3, 8, 36, 3])
