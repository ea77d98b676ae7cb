"""ResNet-9/18 with per-layer passport flags -- drop-in for the reference's
models/resnet_passport.py:20-188 (scheme V1) and, through the `passport_cls` hook, for
models/resnet_passport_private.py (V2/V3, see resnet_passport_private.py here).

Module names (convbnrelu_1, layerN.M.{convbnrelu_1,convbn_2,shortcut}, linear) and hence every
state_dict key equal the reference's.  Every conv block of a BasicBlock is built with relu=True,
including convbn_2 and the shortcut, so the ReLU is applied before and after the residual add
(models/resnet_passport.py:26-30,84).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from deepipr_amd import cuts
from deepipr_amd.models._builders import ind_matters, shared_trunk, trunk_sharing_enabled, PASSPORT_TYPES, conv_factory, run_layer, run_layer_tail
from deepipr_amd.models.layers.conv2d import dual_tail
from deepipr_amd.models.layers.passportconv2d import PassportBlock
from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
from deepipr_amd.passport_ops import conv2d, gamma_beta_batch, max_pool, pooled_linear, stage_groups, with_wino_weights


_SHARED_CONV = os.environ.get('DEEPIPR_NO_SHARED_CONV') != '1'      # read once, at import (A/B switch)
_STACKED = os.environ.get('DEEPIPR_NO_STACKED_BRANCHES') != '1'     # lockstep branches behind the split (StackShare)


def lockstep_pair(layer, inp, stacked, residual=None, residual_stacked=True):
    """A private passport layer for BOTH branches of a dual forward at once (passport_ops.StackShare).
    inp: the layer's input -- the [2N] stack of the branches' inputs (`stacked`), or the one [N] tensor both branches see
    (the layers right behind the point where they part).  ONE convolution either way (forward, backward-data, and a weight
    gradient that carries the private branch's rank-2 term); the norm + affine (+ residual tail) kernels run per branch --
    public first: the reference's order of norm updates -- and write the halves of one buffer.
    -> the [2N] stack of the outputs; with `residual` (stacked, or shared by the branches) the pair of handles of it."""
    from deepipr_amd import passport_ops as P
    n = inp.shape[0] // 2 if stacked else inp.shape[0]
    share = P.StackShare(n)
    conv = P.conv2d(layer.conv, inp, share=share)
    c0, c1 = P.unstack(conv) if stacked else (conv, conv)
    if residual is None:
        r0 = r1 = None
    else:
        r0, r1 = P.unstack(residual) if residual_stacked else (residual, residual)
    xin = inp.detach()[:n]                                   # geometry only: the layer is handed its convolution
    outs = [layer(xin, False, b, _residual=r, _conv_out=c, _stack=(share, b, stacked))
            for b, (c, r) in enumerate(((c0, r0), (c1, r1)))]
    if residual is None:
        return P.restack(outs[0], outs[1])
    return P.restack(outs[0][0], outs[1][0]), P.restack(outs[0][1], outs[1][1])


class BasicPassportBlock(nn.Module):
    expansion = 1
    passport_cls = PassportBlock

    def __init__(self, in_planes, planes, stride=1, passport_kwargs={}):
        super().__init__()
        cls = self.passport_cls
        self.convbnrelu_1 = conv_factory(passport_kwargs['convbnrelu_1'], cls)(in_planes, planes, 3, stride, 1)
        self.convbn_2 = conv_factory(passport_kwargs['convbn_2'], cls)(planes, planes, 3, 1, 1)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = conv_factory(passport_kwargs['shortcut'], cls)(in_planes, self.expansion * planes,
                                                                          1, stride, 0)

    def has_projection(self):
        return not isinstance(self.shortcut, nn.Sequential)

    def set_intermediate_keys(self, pretrained_block, x, y=None):
        """Hand each passport layer the activations that feed it in the pretrained plain block
        (models/resnet_passport.py:32-65); returns the block's outputs for (x, y)."""
        def both(fn, a, b):
            return fn(a), (fn(b) if b is not None else None)

        if isinstance(self.convbnrelu_1, PASSPORT_TYPES):
            self.convbnrelu_1.set_key(x, y)
        out_x, out_y = both(pretrained_block.convbnrelu_1, x, y)
        if isinstance(self.convbn_2, PASSPORT_TYPES):
            self.convbn_2.set_key(out_x, out_y)
        out_x, out_y = both(pretrained_block.convbn_2, out_x, out_y)
        if self.has_projection():
            if isinstance(self.shortcut, PASSPORT_TYPES):
                self.shortcut.set_key(x, y)
            sc_x, sc_y = both(pretrained_block.shortcut, x, y)
        else:
            sc_x, sc_y = x, y
        out_x = F.relu(out_x + sc_x)
        if y is not None:
            out_y = F.relu(out_y + sc_y)
        return out_x, out_y

    def shared_convs(self, x, skip):
        """-> (conv of convbnrelu_1, conv of the shortcut) for the layers whose data convolution two calls of this
        block on the SAME input may share (private passport layers: the two branches of a V2 / V3 dual forward differ
        only from the affine on), None for the others."""
        def one(layer, inp):
            if isinstance(layer, PassportPrivateBlock) and layer.shareable_conv(inp):
                return conv2d(layer.conv, inp)
            return None
        return one(self.convbnrelu_1, x), (one(self.shortcut, skip) if self.has_projection() else None)

    def forward_pair(self, x, skip, force_passport=False, ind=0, preconv=(None, None)):
        """The block on (x, skip) = two handles of the same input -- one per consumer, so that the previous block's
        tail sees their gradients separately (passport_ops.add_relu_fork) -- returning two handles of the output.
        preconv: shared_convs(x, skip), computed once by a caller that runs the block twice on this input."""
        if preconv[0] is not None:
            out = self.convbnrelu_1(x, force_passport, ind, _conv_out=preconv[0])
        else:
            out = run_layer(self.convbnrelu_1, x, force_passport, ind)
        if isinstance(self.convbn_2, PASSPORT_TYPES):
            self.convbn_2.ensure_key(out)                # lazily drawn random keys: the reference's layer order
        if self.has_projection():
            pair = dual_tail(self.convbn_2, self.shortcut, out, skip)     # both plain ConvBlocks: one launch for the two
            if pair is not None:                                          # norm layers and the tail
                return pair
        if preconv[1] is not None:
            sc = self.shortcut(skip, force_passport, ind, _conv_out=preconv[1])
        else:
            sc = run_layer(self.shortcut, skip, force_passport, ind) if self.has_projection() else skip
        # convbn_2, + shortcut, ReLU: folded into convbn_2's own norm kernels when they take the single-pass form
        return run_layer_tail(self.convbn_2, out, sc, force_passport, ind)

    def lockstep_ok(self, x):
        """Both branches of a dual forward can run this block in lockstep (lockstep_pair) on inputs shaped like x."""
        layers = [self.convbnrelu_1, self.convbn_2] + ([self.shortcut] if self.has_projection() else [])
        if not all(isinstance(m, PassportPrivateBlock) for m in layers):
            return False
        return all(m.stackable(x) for m in layers)           # (x stands for every layer's input: device / dtype / rank only)

    def forward_pair_dual(self, x, skip, stacked):
        """forward_pair for both branches at once: (x, skip) are the [2N] stacks of the branches' inputs, or -- in the
        block where the branches part -- the two handles of the one input they share.  -> two handles of the [2N] stack."""
        out = lockstep_pair(self.convbnrelu_1, x, stacked)
        n = out.shape[0] // 2
        self.convbn_2.ensure_key(out.detach()[:n])           # lazily drawn random keys: the reference's layer order
        if self.has_projection():
            return lockstep_pair(self.convbn_2, out, True, lockstep_pair(self.shortcut, skip, stacked), True)
        return lockstep_pair(self.convbn_2, out, True, skip, stacked)

    def forward(self, x, force_passport=False, ind=0):
        return self.forward_pair(x, x, force_passport, ind)[0]


class ResNetPassport(nn.Module):
    def __init__(self, block, num_blocks, num_classes=10, passport_kwargs={}, pretrained=False, imagenet=False):
        super().__init__()
        if pretrained and num_classes == 1000:
            raise NotImplementedError('torchvision-pretrained ImageNet weights are not available offline; '
                                      'load a state_dict instead')
        self.in_planes = 64
        self.num_blocks = num_blocks
        stem = conv_factory(passport_kwargs['convbnrelu_1'], block.passport_cls)
        if num_classes == 1000 or imagenet:                    # 224x224 stem: 7x7/2 conv + 3x3/2 max-pool
            self.convbnrelu_1 = nn.Sequential(stem(3, 64, 7, 2, 3), nn.MaxPool2d(3, 2, 1))
        else:                                                  # CIFAR stem
            self.convbnrelu_1 = stem(3, 64, 3, 1, 1)
        self.layer1 = self._make_layer(block, 64, num_blocks[0], 1, passport_kwargs['layer1'])
        self.layer2 = self._make_layer(block, 128, num_blocks[1], 2, passport_kwargs['layer2'])
        self.layer3 = self._make_layer(block, 256, num_blocks[2], 2, passport_kwargs['layer3'])
        self.layer4 = self._make_layer(block, 512, num_blocks[3], 2, passport_kwargs['layer4'])
        self.linear = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, planes, num_blocks, stride, passport_kwargs):
        layers = []
        for i, s in enumerate([stride] + [1] * (num_blocks - 1)):
            layers.append(block(self.in_planes, planes, s, passport_kwargs[str(i)]))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    def _stem(self, x, force_passport, ind):
        """-> two handles (out, skip) of the stem's output: it feeds layer1's first conv AND its identity shortcut."""
        from deepipr_amd.models.layers.conv2d import ConvBlock
        if isinstance(self.convbnrelu_1, nn.Sequential):
            y = max_pool(self.convbnrelu_1[1], run_layer(self.convbnrelu_1[0], x, force_passport, ind))
            return y, y
        if isinstance(self.convbnrelu_1, ConvBlock):
            return self.convbnrelu_1.forward_fork(x)
        y = run_layer(self.convbnrelu_1, x, force_passport, ind)
        return y, y

    def set_intermediate_keys(self, pretrained_model, x, y=None):
        """models/resnet_passport.py:145-161."""
        with torch.no_grad():
            stem = self.convbnrelu_1[0] if isinstance(self.convbnrelu_1, nn.Sequential) else self.convbnrelu_1
            if isinstance(stem, PASSPORT_TYPES):
                stem.set_key(x, y)
            x = pretrained_model.convbnrelu_1(x)
            if y is not None:
                y = pretrained_model.convbnrelu_1(y)
            for name in ('layer1', 'layer2', 'layer3', 'layer4'):
                for mine, theirs in zip(getattr(self, name), getattr(pretrained_model, name)):
                    x, y = mine.set_intermediate_keys(theirs, x, y)

    def backward_stages(self):
        """Stages of the staged data-parallel backward (experiments/staged.py), last layers first:
        [(cut name = the activation that bounds the stage from below, modules whose gradients are complete once the
        stage has run)].  75 % of a ResNet18's parameter bytes sit in layer4, 19 % in layer3, 6 % in everything before:
        the gradient buckets of flat_sgd.py follow these stages -- few, large messages (xGMI is point-to-point and
        per-link bound); layer3 is a stage of its own because the layers before it run split-channel kernels, which the
        "exclusive" exchange policy keeps clear of collectives (staged.py)."""
        return [('layer4.0', [self.layer4, self.linear]), ('layer3.0', [self.layer3]),
                (None, [self.convbnrelu_1, self.layer1, self.layer2])]

    def passport_layers(self):
        return [m for m in self.modules() if isinstance(m, PASSPORT_TYPES)]

    def _blocks(self):
        return [(li, bi, block) for li, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4))
                for bi, block in enumerate(layer)]

    def _run_blocks(self, out, skip, blocks, force_passport, ind, marked=False, preconv=None):
        for i, (li, bi, block) in enumerate(blocks):
            if li >= 2 and bi == 0 and not (marked and i == 0):      # the cut points backward_stages() names
                out, skip = cuts.mark('layer%d.%d' % (li + 1, bi), out, skip)
            if i == 0 and preconv is not None:
                out, skip = block.forward_pair(out, skip, force_passport, ind, preconv)
            else:
                out, skip = block.forward_pair(out, skip, force_passport, ind)
        return out

    @staticmethod
    def _lockstep_ok(post, x):
        """Every block behind the split can run both branches in lockstep (BasicPassportBlock.lockstep_ok)."""
        return all(hasattr(block, 'lockstep_ok') and block.lockstep_ok(x) for _li, _bi, block in post)

    def _head(self, out):
        # avg-pool to 1x1 + Linear: one launch per direction on the CIFAR-geometry heads (passport_ops.pooled_linear)
        return pooled_linear(self.linear, out)

    @with_wino_weights
    def forward(self, x, force_passport=False, ind=0):
        # gamma / beta of all passport layers in one GEMV launch, up front (they depend on weights and keys only)
        with gamma_beta_batch(self.passport_layers() if x.is_cuda else (), force_passport, ind, stage_groups(self)):
            out, skip = self._stem(x, force_passport, ind)
            out = self._run_blocks(out, skip, self._blocks(), force_passport, ind)
        return self._head(out)

    @with_wino_weights
    def forward_dual(self, x, force_passport=False):
        """-> (self(x, ind=0), self(x, ind=1)), the two forward passes of a V2 / V3 step (trainer_private.py:159-171),
        with everything in front of the first private passport layer run ONCE (_builders.shared_trunk)."""
        blocks = self._blocks()
        split = next((i for i, (_l, _b, blk) in enumerate(blocks) if ind_matters(blk)), len(blocks))
        trunk = [self.convbnrelu_1] + [blk for _l, _b, blk in blocks[:split]]
        if (ind_matters(self.convbnrelu_1) or split == 0 or not trunk_sharing_enabled()
                or not shared_trunk.possible(trunk)):
            return self.forward(x, force_passport, 0), self.forward(x, force_passport, 1)   # hooks: the caller fires them
        with shared_trunk(trunk):
            out, skip = self._stem(x, force_passport, 0)
            for li, bi, block in blocks[:split]:
                if li >= 2 and bi == 0:
                    out, skip = cuts.mark('layer%d.%d' % (li + 1, bi), out, skip)
                out, skip = block.forward_pair(out, skip, force_passport, 0)
        if split < len(blocks) and blocks[split][0] >= 2 and blocks[split][1] == 0:
            # a cut right where the branches part: ONE pair of leaves for both branches (their gradients meet there in
            # the order of the un-cut backward pass)
            out, skip = cuts.mark('layer%d.%d' % (blocks[split][0] + 1, 0), out, skip)
        post = blocks[split:]
        if (_STACKED and not force_passport and x.is_cuda and post and self._lockstep_ok(post, out)):
            # Behind the split the branches run in LOCKSTEP on halves of one buffer: every private passport layer convolves
            # the 2N-image stack once (forward, backward-data, weight gradient) and only the norm + affine kernels run per
            # branch (lockstep_pair; DEEPIPR_NO_STACKED_BRANCHES=1 restores one pass per branch)
            from deepipr_amd import passport_ops as P
            with gamma_beta_batch(self.passport_layers(), False, 1, stage_groups(self)):
                stacked = False
                for i, (li, bi, block) in enumerate(post):
                    if li >= 2 and bi == 0 and i > 0:
                        out, skip = cuts.mark('layer%d.%d' % (li + 1, bi), out, skip)
                    out, skip = block.forward_pair_dual(out, skip, stacked)
                    stacked = True
            return tuple(P.unstack(self._head(out)))
        # the first layers behind the split see the same input in both branches and convolve it with the same weight:
        # that convolution (and its backward: one pass with the branches' summed gradient) is shared as well
        preconv = None
        if split < len(blocks) and hasattr(blocks[split][2], 'shared_convs') and _SHARED_CONV:
            preconv = blocks[split][2].shared_convs(out, skip)
            if preconv[0] is None and preconv[1] is None:
                preconv = None
        outs = []
        for ind in (0, 1):                               # public branch first: the reference's order of norm updates
            with gamma_beta_batch(self.passport_layers() if x.is_cuda else (), force_passport, ind, stage_groups(self)):
                outs.append(self._head(self._run_blocks(out, skip, blocks[split:], force_passport, ind, marked=True,
                                                        preconv=preconv)))
        return outs[0], outs[1]


def ResNet18Passport(**model_kwargs):
    return ResNetPassport(BasicPassportBlock, [2, 2, 2, 2], **model_kwargs)


def ResNet9Passport(**model_kwargs):
    return ResNetPassport(BasicPassportBlock, [1, 1, 1, 1], **model_kwargs)


class BottleneckPassportBlock(nn.Module):
    """Bottleneck (1x1 -> 3x3 -> 1x1, expansion 4) with per-conv passport flags, for BASELINE.json's ResNet50
    configuration.  The reference has NO passport bottleneck (models/resnet_passport.py:183-188 only builds
    BasicPassportBlock); this block is composed from its own pieces the way its plain Bottleneck
    (models/resnet_normal.py:30-49) is: convbnrelu_1 / convbnrelu_2 with ReLU, convbn_3 and the projection
    shortcut WITHOUT ReLU, ReLU after the residual add.  There is therefore no reference oracle for it; it is
    pinned against the oracle's PassportLayerRef composed the same way (tests/test_parity_gpu.py)."""
    expansion = 4
    passport_cls = PassportBlock

    def __init__(self, in_planes, planes, stride=1, passport_kwargs={}):
        super().__init__()

        def make(kw, i, o, ks, s, pd, relu):
            if kw['flag']:
                return self.passport_cls(i, o, ks, s, pd, passport_kwargs=kw, relu=relu)
            from deepipr_amd.models.layers.conv2d import ConvBlock
            return ConvBlock(i, o, ks, s, pd, bn=kw['norm_type'], relu=relu)
        self.convbnrelu_1 = make(passport_kwargs['convbnrelu_1'], in_planes, planes, 1, 1, 0, True)
        self.convbnrelu_2 = make(passport_kwargs['convbnrelu_2'], planes, planes, 3, stride, 1, True)
        self.convbn_3 = make(passport_kwargs['convbn_3'], planes, self.expansion * planes, 1, 1, 0, False)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != self.expansion * planes:
            self.shortcut = make(passport_kwargs['shortcut'], in_planes, self.expansion * planes, 1, stride, 0, False)

    def has_projection(self):
        return not isinstance(self.shortcut, nn.Sequential)

    def set_intermediate_keys(self, pretrained_block, x, y=None):
        def both(fn, a, b):
            return fn(a), (fn(b) if b is not None else None)
        out_x, out_y = x, y
        for name in ('convbnrelu_1', 'convbnrelu_2', 'convbn_3'):
            layer = getattr(self, name)
            if isinstance(layer, PASSPORT_TYPES):
                layer.set_key(out_x, out_y)
            out_x, out_y = both(getattr(pretrained_block, name), out_x, out_y)
        if self.has_projection():
            if isinstance(self.shortcut, PASSPORT_TYPES):
                self.shortcut.set_key(x, y)
            sc_x, sc_y = both(pretrained_block.shortcut, x, y)
        else:
            sc_x, sc_y = x, y
        return F.relu(out_x + sc_x), (F.relu(out_y + sc_y) if y is not None else None)

    def forward_pair(self, x, skip, force_passport=False, ind=0):
        out = run_layer(self.convbnrelu_1, x, force_passport, ind)
        out = run_layer(self.convbnrelu_2, out, force_passport, ind)
        if isinstance(self.convbn_3, PASSPORT_TYPES):
            self.convbn_3.ensure_key(out)
        if self.has_projection():
            pair = dual_tail(self.convbn_3, self.shortcut, out, skip)     # both plain ConvBlocks: one launch for the two
            if pair is not None:                                          # norm layers and the tail
                return pair
        sc = run_layer(self.shortcut, skip, force_passport, ind) if self.has_projection() else skip
        return run_layer_tail(self.convbn_3, out, sc, force_passport, ind)

    def forward(self, x, force_passport=False, ind=0):
        return self.forward_pair(x, x, force_passport, ind)[0]


def ResNet50Passport(**model_kwargs):
    """ResNet-50 (Bottleneck [3, 4, 6, 3]) with passport flags per conv: `passport_configs/resnet50_passport.json`
    switches on layer4 (9 convs + the projection shortcut)."""
    return ResNetPassport(BottleneckPassportBlock, [3, 4, 6, 3], **model_kwargs)
