"""Plain AlexNet (reference models/alexnet_normal.py:7-86): baseline and key-propagation net."""
import torch.nn as nn

from deepipr_amd.models.layers.conv2d import ConvBlock


class AlexNetNormal(nn.Module):
    def __init__(self, in_channels, num_classes, norm_type='bn', pretrained=False, imagenet=False):
        super().__init__()
        if pretrained:
            raise NotImplementedError('torchvision-pretrained ImageNet weights are not available offline')
        if num_classes == 1000 or imagenet:
            self.features = nn.Sequential(
                ConvBlock(3, 64, 11, 4, 2, bn=norm_type), nn.MaxPool2d(kernel_size=3, stride=2),
                ConvBlock(64, 192, 5, 1, 2, bn=norm_type), nn.MaxPool2d(kernel_size=3, stride=2),
                ConvBlock(192, 384, 3, 1, 1, bn=norm_type),
                ConvBlock(384, 256, 3, 1, 1, bn=norm_type),
                ConvBlock(256, 256, 3, 1, 1, bn=norm_type), nn.MaxPool2d(kernel_size=3, stride=2),
                nn.AdaptiveAvgPool2d((6, 6)))
            self.classifier = nn.Sequential(
                nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
                nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
                nn.Linear(4096, num_classes))
        else:
            self.features = nn.Sequential(
                ConvBlock(in_channels, 64, 5, 1, 2, bn=norm_type), nn.MaxPool2d(kernel_size=2, stride=2),
                ConvBlock(64, 192, 5, 1, 2, bn=norm_type), nn.MaxPool2d(kernel_size=2, stride=2),
                ConvBlock(192, 384, bn=norm_type),
                ConvBlock(384, 256, bn=norm_type),
                ConvBlock(256, 256, bn=norm_type), nn.MaxPool2d(kernel_size=2, stride=2))
            self.classifier = nn.Linear(4 * 4 * 256, num_classes)

    def forward(self, x):
        x = self.features(x)
        return self.classifier(x.view(x.size(0), -1))
