"""SignLoss -- same interface as the reference's models/losses/sign_loss.py:6-59, computed by the HIP
sign-loss kernel (deepipr_sign_loss_fwd/bwd) instead of ~15 tiny ATen ops."""
import torch.nn as nn

from deepipr_amd import passport_ops as P


def _plus(cur, new):
    """cur + new, without a kernel launch for the first term after reset() (0 + x is x): `loss += ...` on the freshly
    reset meters cost two tiny ATen kernels per passport layer and step (models/losses/sign_loss.py:52-54)."""
    if isinstance(cur, (int, float)) and cur == 0:
        return new
    return cur + new


class SignLoss(nn.Module):
    def __init__(self, alpha, b=None):
        super().__init__()
        self.alpha = alpha
        self.register_buffer('b', b)
        self.loss = 0
        self.acc = 0
        self.scale_cache = None

    def set_b(self, b):
        self.b.copy_(b)

    def _need_cache(self):
        if self.scale_cache is None:
            raise Exception('scale_cache is None')
        return self.scale_cache.reshape(-1)

    def get_acc(self):
        """mean(sign(b) == sign(gamma)) -- sign_loss.py:18-23."""
        return P.sign_loss(self._need_cache(), self.b, self.alpha, 0.0)[1]

    def get_loss(self):
        """Hinge part only, sum(alpha*relu(-b*gamma + 0.1)) -- sign_loss.py:25-30."""
        return P.sign_loss(self._need_cache(), self.b, self.alpha, 0.0)[0]

    def add(self, scale):
        """loss += hinge + 1e-5*sum(gamma^2); acc += sign accuracy -- sign_loss.py:32-54 (one launch)."""
        self.scale_cache = scale
        loss, acc, _ = P.sign_loss(scale.reshape(-1), self.b, self.alpha, P.L2)
        self.loss = _plus(self.loss, loss)
        self.acc = _plus(self.acc, acc)

    def add_fused(self, scale, loss, acc):
        """Account for values the fused passport-layer launch already produced."""
        self.scale_cache = scale
        self.loss = _plus(self.loss, loss)
        self.acc = _plus(self.acc, acc)

    def reset(self):
        self.loss = 0
        self.acc = 0
        self.scale_cache = None
