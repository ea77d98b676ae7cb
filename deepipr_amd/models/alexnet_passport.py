"""AlexNet with per-layer passport flags -- drop-in for the reference's models/alexnet_passport.py:9-122
(V1) and, via `passport_cls`, models/alexnet_passport_private.py:9-121 (V2/V3).

CIFAR geometry (num_classes != 1000): 5x5 stem, 2x2 max-pools after features 0, 2 and 6, a single
Linear(4*4*256, num_classes); ImageNet geometry: 11x11/4 stem, 3x3/2 pools, adaptive 6x6 pool and the
three-layer dropout classifier.  `features` indices 0,2,4,5,6 are the conv layers, as in the reference.
"""
import os

import torch
import torch.nn as nn

from deepipr_amd import cuts
from deepipr_amd.models._builders import (PASSPORT_TYPES, conv_factory, ind_matters, run_layer, shared_trunk,
                                          trunk_sharing_enabled)
from deepipr_amd.models.layers.passportconv2d import PassportBlock
from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
from deepipr_amd.passport_ops import conv2d, gamma_beta_batch, stage_groups, with_wino_weights

_CUT = 5            # features[_CUT]'s input is the activation backward_stages() cuts the staged backward at
_SHARED_CONV = os.environ.get('DEEPIPR_NO_SHARED_CONV') != '1'      # read once, at import (A/B switch)

_WIDTHS = {0: 64, 2: 192, 4: 384, 5: 256, 6: 256}
_POOL_AT = (1, 3, 7)


class AlexNetPassport(nn.Module):
    passport_cls = PassportBlock

    def __init__(self, in_channels, num_classes, passport_kwargs, pretrained=False, imagenet=False):
        super().__init__()
        big = num_classes == 1000
        if pretrained and big:
            raise NotImplementedError('torchvision-pretrained ImageNet weights are not available offline; '
                                      'load a state_dict instead')
        geometry = {0: (11, 4, 2) if big else (5, 1, 2), 2: (5, 1, 2), 4: (3, 1, 1), 5: (3, 1, 1), 6: (3, 1, 1)}
        layers, inp = [], in_channels
        for idx in range(8):
            if idx in _POOL_AT:
                layers.append(nn.MaxPool2d(3 if big else 2, 2))
                continue
            k, s, p = geometry[idx]
            layers.append(conv_factory(passport_kwargs[str(idx)], self.passport_cls)(inp, _WIDTHS[idx], k, s, p))
            inp = _WIDTHS[idx]
        if big or imagenet:
            layers.append(nn.AdaptiveAvgPool2d((6, 6)))
        self.features = nn.Sequential(*layers)
        if big or imagenet:
            self.classifier = nn.Sequential(
                nn.Dropout(), nn.Linear(256 * 6 * 6, 4096), nn.ReLU(inplace=True),
                nn.Dropout(), nn.Linear(4096, 4096), nn.ReLU(inplace=True),
                nn.Linear(4096, num_classes))
        else:
            self.classifier = nn.Linear(4 * 4 * 256, num_classes)

    def set_intermediate_keys(self, pretrained_model, x, y=None):
        """models/alexnet_passport.py:104-112."""
        with torch.no_grad():
            for theirs, mine in zip(pretrained_model.features, self.features):
                if isinstance(mine, PASSPORT_TYPES):
                    mine.set_key(x, y)
                x = theirs(x)
                if y is not None:
                    y = theirs(y)

    def backward_stages(self):
        """Stages of the staged data-parallel backward (experiments/staged.py), last layers first: the classifier and
        features 5-6 (the larger half of the CIFAR net's parameters), then everything before."""
        return [('features.%d' % _CUT, list(self.features[_CUT:]) + [self.classifier]), (None, list(self.features[:_CUT]))]

    def _run_features(self, x, lo, hi, force_passport, ind):
        for i in range(lo, hi):
            if i == _CUT:
                x = cuts.mark('features.%d' % _CUT, x)
            x = run_layer(self.features[i], x, force_passport, ind)
        return x

    @with_wino_weights
    def forward(self, x, force_passport=False, ind=0):
        layers = [m for m in self.features if isinstance(m, PASSPORT_TYPES)] if x.is_cuda else ()
        with gamma_beta_batch(layers, force_passport, ind, stage_groups(self)):     # all passport layers' gamma / beta in one GEMV launch
            x = self._run_features(x, 0, len(self.features), force_passport, ind)
        return self.classifier(x.view(x.size(0), -1))

    def _lockstep_ok(self, split, x, force_passport):
        """The layers behind the split are private passport layers that can run both branches in lockstep
        (PassportLayerBase.stackable) and per-sample layers (pooling); DEEPIPR_NO_STACKED_BRANCHES=1 switches it off."""
        from deepipr_amd.models import resnet_passport as R
        if not (R._STACKED and not force_passport and x.is_cuda and split < len(self.features)):
            return False
        if not isinstance(self.features[split], PassportPrivateBlock):
            return False
        for m in list(self.features)[split:]:
            if isinstance(m, PassportPrivateBlock):
                if not m.stackable(x):
                    return False
            elif not isinstance(m, (torch.nn.MaxPool2d, torch.nn.AvgPool2d, torch.nn.Identity)):
                return False
        return True

    @with_wino_weights
    def forward_dual(self, x, force_passport=False):
        """-> (self(x, ind=0), self(x, ind=1)), the two forward passes of a V2 / V3 step (trainer_private.py:159-171),
        with the layers in front of the first private passport layer run ONCE (_builders.shared_trunk)."""
        n = len(self.features)
        split = next((i for i, m in enumerate(self.features) if ind_matters(m)), n)
        trunk = list(self.features[:split])
        if split == 0 or not trunk_sharing_enabled() or not shared_trunk.possible(trunk):
            return self.forward(x, force_passport, 0), self.forward(x, force_passport, 1)   # hooks: the caller fires them
        with shared_trunk(trunk):
            x = self._run_features(x, 0, split, force_passport, 0)
        layers = [m for m in self.features if isinstance(m, PASSPORT_TYPES)] if x.is_cuda else ()
        if self._lockstep_ok(split, x, force_passport):
            # behind the split the branches run in LOCKSTEP on halves of one buffer: one convolution of the 2N-image stack
            # per private passport layer, norm + affine per branch (resnet_passport.lockstep_pair)
            from deepipr_amd import passport_ops as P
            from deepipr_amd.models.resnet_passport import lockstep_pair
            with gamma_beta_batch(layers, False, 1, stage_groups(self)):
                stacked = False
                for i in range(split, n):
                    if i == _CUT:
                        x = cuts.mark('features.%d' % _CUT, x)
                    if isinstance(self.features[i], PassportPrivateBlock):
                        x, stacked = lockstep_pair(self.features[i], x, stacked), True
                    else:
                        x = self.features[i](x)               # per-sample layers (pooling): the stack as it is
            return tuple(P.unstack(self.classifier(x.view(x.size(0), -1))))
        # the first layer behind the split convolves the same input with the same weight in both branches: shared too
        first, conv_out = self.features[split] if split < n else None, None
        # ... unless the branches part exactly at the staged backward's cut (features[_CUT]): _run_features marks the cut on
        # the layer's INPUT, per branch; a convolution computed here, in front of that mark, would tie the two stages
        # together through its autograd node.  The layer then convolves once per branch, as without sharing
        # (tests/test_host_logic.py::test_alexnet_split_at_the_stage_cut_does_not_share_the_convolution).
        if (isinstance(first, PassportPrivateBlock) and first.shareable_conv(x) and split != _CUT and _SHARED_CONV):
            conv_out = conv2d(first.conv, x)
        outs = []
        for ind in (0, 1):                               # public branch first: the reference's order of norm updates
            with gamma_beta_batch(layers, force_passport, ind, stage_groups(self)):
                if conv_out is not None:
                    y = first(x, force_passport, ind, _conv_out=conv_out)
                    y = self._run_features(y, split + 1, n, force_passport, ind)
                else:
                    y = self._run_features(x, split, n, force_passport, ind)
            outs.append(self.classifier(y.view(y.size(0), -1)))
        return outs[0], outs[1]
