"""PassportPrivateBlock -- drop-in for the reference's models/layers/passportconv2d_private.py:11-219
(schemes V2 / V3): ind=0 is the public branch with learnable `scale` / `bias`, ind=1 (or
force_passport) the private branch whose gamma / beta come from `skey_private` / `key_private` and
feed `sign_loss_private`.  Always has a ReLU (:66)."""
from deepipr_amd.models.layers._passport_base import PassportLayerBase


class PassportPrivateBlock(PassportLayerBase):
    KEY, SKEY, SIGN = 'key_private', 'skey_private', 'sign_loss_private'

    def __init__(self, i, o, ks=3, s=1, pd=1, passport_kwargs={}):
        super().__init__()
        self.norm_type = passport_kwargs.get('norm_type', 'bn')
        self.init_public_bit = passport_kwargs.get('init_public_bit', True)     # stored, unused (:44)
        self._build(i, o, ks, s, pd, passport_kwargs, True, learnable_affine=True, always_sign_loss=True)

    def forward(self, x, force_passport=False, ind=0, _residual=None, _conv_out=None, _stack=None):
        return self._forward(x, force_passport, ind, _residual, _conv_out, _stack)

    def forward_tail(self, x, residual, force_passport=False, ind=0):
        """-> two handles of relu(self(x) + residual): this layer as the last one of a residual block."""
        return self(x, force_passport, ind, _residual=residual)
