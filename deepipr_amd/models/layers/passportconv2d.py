"""PassportBlock -- drop-in for the reference's models/layers/passportconv2d.py:11-223 (scheme V1).

    conv -> norm(affine=False) -> gamma * x + beta -> ReLU
    gamma = mean_{b,h,w} conv(skey, W),  beta = mean_{b,h,w} conv(key, W),  sign loss on gamma

Same constructor, attributes (conv, weight, alpha, b, key, skey, scale, bias, bn, relu, sign_loss,
key_type, requires_reset_key), methods and state_dict names as the reference; the arithmetic after
the norm runs in the HIP kernels (see _passport_base.py).
"""
from deepipr_amd.models.layers._passport_base import PassportLayerBase


class PassportBlock(PassportLayerBase):
    KEY, SKEY, SIGN = 'key', 'skey', 'sign_loss'

    def __init__(self, i, o, ks=3, s=1, pd=1, passport_kwargs={}, relu=True):
        super().__init__()
        self._build(i, o, ks, s, pd, passport_kwargs, relu, learnable_affine=False, always_sign_loss=False)

    def get_scale(self, force_passport=False):
        return super().get_scale(force_passport, 0)

    def get_bias(self, force_passport=False):
        return super().get_bias(force_passport, 0)

    def forward(self, x, force_passport=False, _residual=None):
        return self._forward(x, force_passport, 0, _residual)

    def forward_tail(self, x, residual, force_passport=False):
        """-> two handles of relu(self(x) + residual): this layer as the last one of a residual block."""
        return self(x, force_passport, _residual=residual)
