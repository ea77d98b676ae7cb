"""Shared machinery of PassportBlock (V1) and PassportPrivateBlock (V2/V3).

The reference keeps two near-identical 220-line classes (models/layers/passportconv2d.py,
models/layers/passportconv2d_private.py); here the behaviour lives once and the two public classes
only choose buffer names and defaults.  Data conv and norm stay on MIOpen; everything after the norm
(passport conv -> pool -> gamma/beta, sign loss, affine, ReLU and all of their backward) goes through
deepipr_amd.passport_ops, i.e. the HIP kernels.
"""
import random

import numpy as np
import torch
import torch.nn as nn

from deepipr_amd import passport_ops as P
from deepipr_amd.models.layers.conv2d import make_norm
from deepipr_amd.models.losses.sign_loss import SignLoss


def signature_vector(spec, channels, random_sign):
    """passport_kwargs['b'] -> +-1 per output channel (passportconv2d.py:25-40).

    int -> constant vector; str -> 8 bits per character, most significant first ('0' -> -1), written over
    the leading channels of a fresh random-sign vector; more than `channels` bits is an error."""
    if isinstance(spec, int):
        return torch.ones(channels) * spec
    if isinstance(spec, str):
        if len(spec) * 8 > channels:
            raise Exception('Too much bit information')
        vec = random_sign()
        bits = ''.join(format(ord(ch), 'b').zfill(8) for ch in spec)
        vec[:len(bits)] = torch.tensor([1.0 if ch == '1' else -1.0 for ch in bits])
        return vec
    return spec


class PassportLayerBase(nn.Module):
    # subclasses set these
    KEY = 'key'
    SKEY = 'skey'
    SIGN = 'sign_loss'

    def _build(self, i, o, ks, s, pd, passport_kwargs, relu, learnable_affine, always_sign_loss):
        if passport_kwargs == {}:
            print('Warning, passport_kwargs is empty')
        # RNG draw order below equals the reference's constructor, so the same torch seed gives the
        # same initial weights and signature bits (passportconv2d.py:18,25,31,88).
        self.conv = nn.Conv2d(i, o, ks, s, pd, bias=False)
        self.key_type = passport_kwargs.get('key_type', 'random')
        self.weight = self.conv.weight
        self.alpha = passport_kwargs.get('sign_loss', 1)

        def random_sign():
            return torch.sign(torch.rand(o) - 0.5)
        b = signature_vector(passport_kwargs.get('b', random_sign()), o, random_sign)
        self.register_buffer('b', b)
        self.requires_reset_key = False
        if always_sign_loss or self.alpha != 0:
            setattr(self, self.SIGN, SignLoss(self.alpha, self.b))
        else:
            setattr(self, self.SIGN, None)
        self.register_buffer(self.KEY, None)
        self.register_buffer(self.SKEY, None)
        self.init_scale(learnable_affine)
        self.init_bias(learnable_affine)
        norm = make_norm(passport_kwargs.get('norm_type', 'bn'), o, affine=False)
        self.bn = norm if norm is not None else nn.Sequential()
        self.relu = nn.ReLU(inplace=True) if relu else None
        self._pooled = P.PooledKeys()
        self.fuse_norm = True          # fold BatchNorm(affine=False) into the passport kernels when possible
        self.fuse_conv = True          # ... and run the data conv inside that node (in-place three-way dW)
        self.reset_parameters()

    # ------------------------------------------------------------------ parameters
    def init_bias(self, force_init=False):
        if force_init:
            self.bias = nn.Parameter(torch.zeros(self.conv.out_channels, device=self.weight.device))
        else:
            self.bias = None

    def init_scale(self, force_init=False):
        if force_init:
            self.scale = nn.Parameter(torch.ones(self.conv.out_channels, device=self.weight.device))
        else:
            self.scale = None

    def reset_parameters(self):
        nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')

    # ------------------------------------------------------------------ keys
    def passport_selection(self, passport_candidates):
        """n candidate activations -> one [1,C,H,W] passport (passportconv2d.py:90-123): for RGB inputs
        one whole image; otherwise channel j comes from candidate j mod n, a not-yet-used channel of it
        drawn with python's `random` (same draw sequence as the reference)."""
        n, c, h, w = passport_candidates.size()
        if c == 3:
            return passport_candidates[random.randint(0, n - 1)].unsqueeze(0)
        flat = passport_candidates.view(n * c, h, w)
        used, picks = set(), []
        for j in range(c):
            base = (j % n) * c
            pick = base + random.randint(0, c - 1)
            while pick in used:
                pick = base + random.randint(0, c - 1)
            used.add(pick)
            picks.append(pick)
        return flat[picks].unsqueeze(0)

    def set_key(self, x, y=None):
        """x -> bias key, y -> scale key (passportconv2d.py:125-137)."""
        if int(x.size(0)) != 1:
            x = self.passport_selection(x)
            if y is not None:
                y = self.passport_selection(y)
        self.register_buffer(self.KEY, x)
        self.register_buffer(self.SKEY, y)
        self._pooled.clear()                       # new key tensors: never trust address + version alone

    def generate_key(self, *shape):
        shape = [1] + list(shape[1:])
        return np.random.uniform(-1.0, 1.0, shape)

    def get_scale_key(self):
        return getattr(self, self.SKEY)

    def get_bias_key(self):
        return getattr(self, self.KEY)

    def _sign(self):
        return getattr(self, self.SIGN)

    def invalidate_key_cache(self):
        """Forget the cached pooled passport means.  Needed only after the key tensors were modified in place
        by something that does not bump their version counter (e.g. a c10d broadcast)."""
        self._pooled.clear()

    def _geometry(self):
        c = self.conv
        if c.groups != 1 or tuple(c.dilation) != (1, 1) or c.stride[0] != c.stride[1] or c.padding[0] != c.padding[1]:
            raise RuntimeError('passport conv must be a plain square-stride, square-padding convolution')
        return c.kernel_size[0], c.kernel_size[1], c.stride[0], c.padding[0]

    def _pooled_means(self):
        skey, key = self.get_scale_key(), self.get_bias_key()
        if skey is None or key is None:
            raise RuntimeError('passport keys are not set (call set_key, or use key_type="random")')
        kh, kw, stride, pad = self._geometry()
        return skey, key, self._pooled.get(skey, key, kh, kw, stride, pad), stride, pad

    # ------------------------------------------------------------------ gamma / beta
    def _use_param(self, param, force_passport, ind):
        return param is not None and not force_passport and ind == 0

    def _passport_gamma_beta(self):
        skey, key, m, stride, pad = self._pooled_means()
        return P.gamma_beta(self.weight, skey, key, m, stride, pad)

    def get_scale(self, force_passport=False, ind=0):
        """[1,C,1,1] gamma: the learnable `scale` on the public branch, otherwise the pooled response of
        the layer's own conv to the scale key, which also (re)sets the sign loss
        (passportconv2d.py:142-158; private :139-156)."""
        if self._use_param(self.scale, force_passport, ind):
            return self.scale.view(1, -1, 1, 1)
        gamma = self._passport_gamma_beta()[0].view(1, -1, 1, 1)
        sl = self._sign()
        if sl is not None:
            sl.reset()
            sl.add(gamma)
        return gamma

    def get_bias(self, force_passport=False, ind=0):
        """[1,C,1,1] beta (passportconv2d.py:163-175; private :161-173)."""
        if self._use_param(self.bias, force_passport, ind):
            return self.bias.view(1, -1, 1, 1)
        return self._passport_gamma_beta()[1].view(1, -1, 1, 1)

    # ------------------------------------------------------------------ checkpoints
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """Pre-size the lazily created tensors so strict loads of reference checkpoints succeed
        (passportconv2d.py:177-196)."""
        dev = self.weight.device
        # (unlike the reference, tensors that already have the right shape are kept, so an optimiser or a
        # captured hipGraph holding them stays valid across a load)
        for name in (self.KEY, self.SKEY):
            if prefix + name in state_dict:
                cur, want = getattr(self, name), state_dict[prefix + name].size()
                if cur is None or cur.size() != want:
                    self.register_buffer(name, torch.empty(want, device=dev))
        for name in ('scale', 'bias'):
            if prefix + name in state_dict:
                cur, want = getattr(self, name), state_dict[prefix + name].size()
                if cur is None or cur.size() != want:
                    setattr(self, name, nn.Parameter(torch.empty(want, device=dev)))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)
        self._pooled.clear()                       # the keys were refilled in place

    # ------------------------------------------------------------------ forward
    def _forward(self, x, force_passport, ind, residual=None, conv_out=None, stack=None):
        """The layer; with `residual` (the shortcut of a residual block whose last layer this is) the pair of handles
        of relu(layer(x) + residual), folded into the layer's own kernels when they take the single-pass form.
        conv_out: self.conv(x), already computed by the caller (the two branches of a V2 / V3 dual forward share the
        data convolution of the first layers behind the point where they part: same input, same weight)."""
        if stack is not None:                        # stacked branches (passport_ops.StackShare): the caller checked stackable()
            y = self._layer(x, force_passport, ind, residual, conv_out, stack)
        else:
            y = self._layer(x, force_passport, ind, residual, conv_out)     # (the signature tests and attack scripts wrap)
        if residual is None or isinstance(y, tuple):
            return y
        return P.add_relu_fork(y, residual)

    def ensure_key(self, x):
        """key_type='random': draw the keys lazily from numpy's global RNG at the first input
        (passportconv2d.py:210-216).  Callable on its own so that a block can keep the reference's draw order
        (convbnrelu_1, convbn_2, shortcut) while evaluating the shortcut before convbn_2."""
        if (self.get_bias_key() is None and self.key_type == 'random') or self.requires_reset_key:
            self.set_key(torch.tensor(self.generate_key(*x.size()), dtype=x.dtype, device=x.device),
                         torch.tensor(self.generate_key(*x.size()), dtype=x.dtype, device=x.device))

    def _conv_inside(self, x):
        """The data convolution can run inside the fused node (passport_ops._PassportBNLayer, `conv`): a plain
        bias-free conv nobody hooked or re-parametrised -- `self.weight` must still BE conv.weight (it is an alias taken
        in __init__, passportconv2d.py:21; after `conv.weight = nn.Parameter(...)` or torch.nn.utils.parametrize in a
        fine-tune / attack script the node would otherwise convolve with the stale tensor) -- and a dense fp32 GPU
        input."""
        c = self.conv
        return (self.fuse_conv and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and c.bias is None
                and self.weight is c._parameters.get('weight')
                and c.groups == 1 and tuple(c.dilation) == (1, 1) and c.padding_mode == 'zeros'
                and not c._forward_hooks and not c._forward_pre_hooks and not c._backward_hooks)

    def _plan(self, x, public):
        """Which kernels take this call: -> (form, conv_inside).
        form 'bn'   BatchNorm(affine=False) folded into the single-pass / three-launch kernels (deepipr_passport_bn_*)
             'gn'   GroupNorm / InstanceNorm folded in (deepipr_passport_gn_*), when the (sample, group) chunk fits
             'none' norm_type 'none': nothing between conv and affine (deepipr_passport_fwd / _bwd)
             'lib'  the library norm, then the unfused affine kernels
        conv_inside: the data convolution runs inside the fused autograd node, so that the shared weight's three-way
        gradient is formed in MIOpen's wgrad buffer (passport branch only; see _conv_inside).  A new fused form is one
        more line here and one more entry in _layer's dispatch."""
        norm = self.bn
        inside = (not public) and self._conv_inside(x)
        if self.fuse_norm and P.bn_is_fusable(norm):
            return 'bn', inside
        if (self.fuse_norm and not getattr(norm, 'affine', False) and P.norm_groups(norm)
                and x.dim() == 4 and P.gn_supported_shape(norm, P.conv_out_shape(x, self.conv))):
            return 'gn', inside
        if isinstance(norm, nn.Sequential) and len(norm) == 0:
            return 'none', inside
        return 'lib', False

    def shareable_conv(self, x):
        """The data convolution may be computed once by the caller and handed to several calls of this layer
        (`conv_out`): a plain call of the conv module nobody hooked."""
        c = self.conv
        return (x.dim() == 4 and not c._forward_hooks and not c._forward_pre_hooks and not c._backward_hooks
                and not self._forward_hooks and not self._forward_pre_hooks)

    def stackable(self, x):
        """The two branches of a dual forward may run this layer in lockstep on halves of one buffer
        (passport_ops.StackShare): both branches take the fused BatchNorm form (learnable scale AND bias on the public one),
        around a plain un-hooked convolution the caller may run once for both."""
        return (self.scale is not None and self.bias is not None and self.fuse_norm and P.bn_is_fusable(self.bn)
                and x.is_cuda and x.dtype == torch.float32 and self.shareable_conv(x) and self._conv_inside(x)
                and P.conv_plain(self.conv, x))      # conv2d(share=...)'s own test: autocast, process-wide module hooks included

    def _layer(self, x, force_passport, ind, residual, conv_out=None, stack=None):
        self.ensure_key(x)
        relu = self.relu is not None
        p_scale = self._use_param(self.scale, force_passport, ind)
        p_bias = self._use_param(self.bias, force_passport, ind)
        if p_scale != p_bias:                        # mixed (only one of scale / bias learnable): compose the operators
            x = self.bn(self.conv(x) if conv_out is None else conv_out)
            return P.affine_relu(x, self.get_scale(force_passport, ind), self.get_bias(force_passport, ind), relu)
        public = p_scale
        form, inside = self._plan(x, public)
        tail = None
        if form == 'bn' and residual is not None and P.bn_tail_fusable(self.bn, P.conv_out_shape(x, self.conv)):
            tail = residual                          # relu(layer + shortcut) folded into the layer's own kernels
        if conv_out is not None:                     # the caller ran the convolution (shared by two calls)
            inside, x = False, conv_out
        elif not inside:
            x = P.conv2d(self.conv, x)
        if public:                                   # learnable scale / bias, no sign loss
            if form == 'bn':
                return P.bn_affine_relu(x, self.scale, self.bias, self.bn, relu, tail, stack=stack)
            if form == 'gn':
                return P.gn_affine_relu(x, self.scale, self.bias, self.bn, relu)
            return P.affine_relu(self.bn(x), self.scale, self.bias, relu)
        # passport branch.  The loss is the SignLoss module's own (its b and alpha, sign_loss.py:27: set_b() and
        # checkpoints count)
        sl = self._sign()
        b, alpha = (sl.b, sl.alpha) if sl is not None else (None, 0.0)
        skey, key, m, stride, pad = self._pooled_means()
        if form == 'bn':
            pre, self._gb_pre = self._gb_pre, None   # gamma / beta from the net's batched GEMV launch, if any
            y, gamma, _beta, loss, acc, _bits = P.passport_bn_layer(
                x, self.weight, skey, key, b, m, self.bn, alpha, relu, stride, pad, tail, conv_inside=inside, pre=pre,
                stack=stack)
        elif form == 'gn':
            y, gamma, _beta, loss, acc, _bits = P.passport_gn_layer(
                x, self.weight, skey, key, b, m, self.bn, alpha, relu, stride, pad, conv_inside=inside)
        else:
            y, gamma, _beta, loss, acc, _bits = P.passport_layer(
                x if (inside or form == 'none') else self.bn(x), self.weight, skey, key, b, m, alpha, relu, stride, pad,
                conv_inside=inside)
        if sl is not None:
            sl.reset()
            sl.add_fused(gamma.view(1, -1, 1, 1), loss, acc)
        return y

    _gb_pre = None

    def batched_gamma_beta_request(self, force_passport, ind):
        """-> (weight, pooled means) when this layer's next forward will take the fused BatchNorm passport form, whose
        gamma / beta the net may then compute for all such layers in ONE launch (passport_ops.gamma_beta_batch);
        None otherwise."""
        if (self._use_param(self.scale, force_passport, ind) or self._use_param(self.bias, force_passport, ind)
                or not (self.fuse_norm and P.bn_is_fusable(self.bn)) or not self.weight.is_cuda
                or self.get_bias_key() is None or self.get_scale_key() is None or self.requires_reset_key):
            return None
        return self.weight, self._pooled_means()[2]
