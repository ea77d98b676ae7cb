"""Layer factories shared by the AlexNet / ResNet passport models."""
from deepipr_amd.models.layers.conv2d import ConvBlock
from deepipr_amd.models.layers.passportconv2d import PassportBlock
from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock

PASSPORT_TYPES = (PassportBlock, PassportPrivateBlock)


def conv_factory(passport_kwargs, passport_cls):
    """passport_kwargs['flag'] picks the passport layer or a plain ConvBlock with the same geometry
    (reference get_convblock, models/resnet_passport.py:10-17)."""
    def make(i, o, ks, s, pd):
        if passport_kwargs['flag']:
            return passport_cls(i, o, ks, s, pd, passport_kwargs=passport_kwargs)
        return ConvBlock(i, o, ks, s, pd, bn=passport_kwargs['norm_type'])
    return make


def run_layer(layer, x, force_passport, ind):
    """Call a layer with the extra arguments its type accepts."""
    if isinstance(layer, PassportPrivateBlock):
        return layer(x, force_passport, ind)
    if isinstance(layer, PassportBlock):
        return layer(x, force_passport)
    return layer(x)


def run_layer_tail(layer, x, residual, force_passport, ind):
    """`layer` as the last layer of a residual block: -> two handles of relu(layer(x) + residual)."""
    if isinstance(layer, PassportPrivateBlock):
        return layer.forward_tail(x, residual, force_passport, ind)
    if isinstance(layer, PassportBlock):
        return layer.forward_tail(x, residual, force_passport)
    return layer.forward_tail(x, residual)
