"""Plain AlexNet: baseline net and key-propagation net (reference models/alexnet_normal.py:7-86).

`features` is assembled from a per-geometry table of (kind, args) entries so that its indices line up with
AlexNetPassport's (convs at 0, 2, 4, 5, 6; pools at 1, 3, 7), which is what set_intermediate_keys zips over.
"""
import torch.nn as nn

from deepipr_amd.models.layers.conv2d import ConvBlock
from deepipr_amd.passport_ops import with_wino_weights

# (out_channels, kernel, stride, padding) of the five conv layers; pools sit after conv 0, 1 and 4
_CIFAR = dict(convs=[(64, 5, 1, 2), (192, 5, 1, 2), (384, 3, 1, 1), (256, 3, 1, 1), (256, 3, 1, 1)], pool=(2, 2))
_IMAGENET = dict(convs=[(64, 11, 4, 2), (192, 5, 1, 2), (384, 3, 1, 1), (256, 3, 1, 1), (256, 3, 1, 1)], pool=(3, 2))
_POOL_AFTER = (0, 1, 4)


def _imagenet_head(num_classes):
    return nn.Sequential(nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
                         nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
                         nn.Linear(4096, num_classes))


class AlexNetNormal(nn.Module):
    def __init__(self, in_channels, num_classes, norm_type='bn', pretrained=False, imagenet=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError('torchvision-pretrained ImageNet weights are not available offline')
        big = imagenet or num_classes == 1000
        spec = _IMAGENET if big else _CIFAR
        layers, width = [], (3 if big else in_channels)
        for i, (out, k, s, p) in enumerate(spec['convs']):
            layers.append(ConvBlock(width, out, k, s, p, bn=norm_type))
            width = out
            if i in _POOL_AFTER:
                layers.append(nn.MaxPool2d(kernel_size=spec['pool'][0], stride=spec['pool'][1]))
        if big:
            layers.append(nn.AdaptiveAvgPool2d((6, 6)))
        self.features = nn.Sequential(*layers)
        self.classifier = _imagenet_head(num_classes) if big else nn.Linear(4 * 4 * width, num_classes)

    @with_wino_weights
    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.flatten(1))
