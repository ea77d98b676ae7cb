"""ResNet-18 with PassportPrivateBlock layers (schemes V2 / V3) -- drop-in for the reference's
models/resnet_passport_private.py:20-186.  forward(x, force_passport=False, ind=0): ind=0 is the
public branch (learnable scale/bias), ind=1 the private passport branch."""
from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
from deepipr_amd.models.resnet_passport import BasicPassportBlock, ResNetPassport


class BasicPrivateBlock(BasicPassportBlock):
    passport_cls = PassportPrivateBlock


class ResNetPrivate(ResNetPassport):
    pass


def ResNet18Private(**model_kwargs):
    return ResNetPrivate(BasicPrivateBlock, [2, 2, 2, 2], **model_kwargs)
