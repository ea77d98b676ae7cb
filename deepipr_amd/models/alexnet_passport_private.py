"""AlexNet with PassportPrivateBlock layers (V2 / V3) -- drop-in for the reference's
models/alexnet_passport_private.py:9-121."""
from deepipr_amd.models.alexnet_passport import AlexNetPassport
from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock


class AlexNetPassportPrivate(AlexNetPassport):
    passport_cls = PassportPrivateBlock
