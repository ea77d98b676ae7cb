"""Plain-PyTorch CPU restatement of the DeepIPR passport nets and train steps.

TEST INFRASTRUCTURE (see oracle/__init__.py): stock ATen ops only, no custom kernels.  Written
fresh from the behaviour of the reference; every piece cites the lines it restates (paths relative
to /root/reference).  Module / parameter / buffer names equal the reference's so that
oracle.patterns.fill_state() gives the reference, this oracle and the HIP build identical weights.
"""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.np_passport import parse_signature


# ----------------------------------------------------------------------------- sign loss
class SignLossRef(nn.Module):
    """models/losses/sign_loss.py:6-59 -- hinge(0.1) * alpha + 1e-5 * L2, sign-match accuracy."""

    def __init__(self, alpha, b):
        super().__init__()
        self.alpha = alpha
        self.register_buffer('b', b)
        self.reset()

    def reset(self):                                           # :56-59
        self.loss, self.acc, self.scale_cache = 0, 0, None

    def add(self, scale):                                      # :32-54
        self.scale_cache = scale
        g, b = scale.view(-1), self.b.view(-1)
        self.loss = self.loss + (self.alpha * F.relu(-b * g + 0.1)).sum()         # :27,52
        self.loss = self.loss + 0.00001 * g.pow(2).sum()                          # :53
        self.acc = self.acc + (torch.sign(b) == torch.sign(g)).float().mean()     # :20,54


def _norm(norm_type, o, affine):
    """models/layers/passportconv2d.py:56-64 (affine=False) and models/layers/conv2d.py:15-22."""
    if norm_type == 'bn':
        return nn.BatchNorm2d(o, affine=affine)
    if norm_type == 'gn':
        return nn.GroupNorm(o // 16, o, affine=affine)
    if norm_type == 'in':
        return nn.InstanceNorm2d(o)      # conv2d.py:19 passes no affine flag: InstanceNorm2d defaults to affine=False
    return None


# ----------------------------------------------------------------------------- blocks
class ConvBlockRef(nn.Module):
    """models/layers/conv2d.py:5-36: conv(+bias iff no norm) -> norm(affine) -> ReLU."""

    def __init__(self, i, o, ks=3, s=1, pd=1, bn='bn', relu=True):
        super().__init__()
        self.conv = nn.Conv2d(i, o, ks, s, pd, bias=bn == 'none')
        self.bn = _norm(bn, o, True)
        self.use_relu = relu
        nn.init.kaiming_normal_(self.conv.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        return F.relu(x) if self.use_relu else x


class PassportLayerRef(nn.Module):
    """PassportBlock (models/layers/passportconv2d.py:11-223) when private=False,
    PassportPrivateBlock (models/layers/passportconv2d_private.py:11-219) when private=True."""

    def __init__(self, i, o, ks=3, s=1, pd=1, passport_kwargs={}, relu=True, private=False):
        super().__init__()
        self.private = private
        self.conv = nn.Conv2d(i, o, ks, s, pd, bias=False)
        self.weight = self.conv.weight                                           # :21 alias
        self.key_type = passport_kwargs.get('key_type', 'random')
        self.alpha = passport_kwargs.get('sign_loss', 1)
        rand = torch.sign(torch.rand(o) - 0.5).numpy()
        b = torch.from_numpy(parse_signature(passport_kwargs.get('b', None), o, rand))   # :25-40
        self.register_buffer('b', b)
        kn = '_private' if private else ''
        self._k, self._sk, self._sl = 'key' + kn, 'skey' + kn, 'sign_loss' + kn
        if private or self.alpha != 0:                                           # :45-48 / private :48
            setattr(self, self._sl, SignLossRef(self.alpha, self.b))
        else:
            self.sign_loss = None
        self.register_buffer(self._k, None)
        self.register_buffer(self._sk, None)
        if private:                                                              # private :53-54
            self.scale = nn.Parameter(torch.ones(o))
            self.bias = nn.Parameter(torch.zeros(o))
        else:
            self.scale = None
            self.bias = None
        self.bn = _norm(passport_kwargs.get('norm_type', 'bn'), o, False)
        self.use_relu = True if private else relu                                # private :66
        nn.init.kaiming_normal_(self.weight, mode='fan_out', nonlinearity='relu')  # :87-88

    @staticmethod
    def passport_selection(cands):                                               # :90-123
        """n candidates -> one passport: an RGB input keeps one whole image; otherwise output channel j is a
        not-yet-taken channel of candidate j mod n, drawn (with redraws on collision) from python's `random`."""
        n, c, h, w = cands.size()
        if c == 3:
            return cands[random.randint(0, n - 1)].unsqueeze(0)
        flat = cands.view(n * c, h, w)
        taken = [False] * (n * c)
        chosen = []
        for j in range(c):
            src = j % n
            idx = src * c + random.randint(0, c - 1)
            while taken[idx]:
                idx = src * c + random.randint(0, c - 1)
            taken[idx] = True
            chosen.append(flat[idx])
        return torch.stack(chosen, dim=0).unsqueeze(0)

    def set_key(self, x, y=None):                                                # :125-137
        if int(x.size(0)) != 1:
            x = self.passport_selection(x)
            if y is not None:
                y = self.passport_selection(y)
        self.register_buffer(self._k, x)
        self.register_buffer(self._sk, y)

    def _pooled(self, key):                                                      # :148-152 / :169-173
        r = self.conv(key)
        b, c = r.size(0), r.size(1)
        return r.view(b, c, -1).mean(dim=2).view(b, c, 1, 1).mean(dim=0).view(1, c, 1, 1)

    def get_scale(self, force_passport=False, ind=0):                            # :142-158 / private :139-156
        if self.scale is not None and not force_passport and ind == 0:
            return self.scale.view(1, -1, 1, 1)
        scale = self._pooled(getattr(self, self._sk))
        sl = getattr(self, self._sl)
        if sl is not None:
            sl.reset()
            sl.add(scale)
        return scale

    def get_bias(self, force_passport=False, ind=0):                             # :163-175 / private :161-173
        if self.bias is not None and not force_passport and ind == 0:
            return self.bias.view(1, -1, 1, 1)
        return self._pooled(getattr(self, self._k))

    def forward(self, x, force_passport=False, ind=0):                           # :209-223 / private :205-219
        if getattr(self, self._k) is None and self.key_type == 'random':
            shape = [1] + list(x.shape[1:])                                      # :198-207
            k = torch.tensor(np.random.uniform(-1.0, 1.0, shape), dtype=x.dtype)
            sk = torch.tensor(np.random.uniform(-1.0, 1.0, shape), dtype=x.dtype)
            self.set_key(k, sk)
        x = self.conv(x)
        if self.bn is not None:
            x = self.bn(x)
        x = self.get_scale(force_passport, ind) * x + self.get_bias(force_passport, ind)
        return F.relu(x) if self.use_relu else x


def _make_block(kw, private):
    """get_convblock: models/resnet_passport.py:10-17, models/resnet_passport_private.py:10-17."""
    def build(*args):
        if kw['flag']:
            return PassportLayerRef(*args, passport_kwargs=kw, private=private)
        return ConvBlockRef(*args, bn=kw['norm_type'])
    return build


def _call(m, x, force_passport, ind):
    if isinstance(m, PassportLayerRef):
        return m(x, force_passport, ind)
    return m(x)


# ----------------------------------------------------------------------------- ResNet
class BasicBlockRef(nn.Module):
    """BasicPassportBlock (models/resnet_passport.py:20-85) / BasicPrivateBlock
    (models/resnet_passport_private.py:20-86).  c2 and shortcut keep relu=True (:27,30), so ReLU is
    applied before and after the residual add."""

    def __init__(self, in_planes, planes, stride, kw, private):
        super().__init__()
        self.convbnrelu_1 = _make_block(kw['convbnrelu_1'], private)(in_planes, planes, 3, stride, 1)
        self.convbn_2 = _make_block(kw['convbn_2'], private)(planes, planes, 3, 1, 1)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != planes:
            self.shortcut = _make_block(kw['shortcut'], private)(in_planes, planes, 1, stride, 0)

    def set_intermediate_keys(self, plain_block, x, y):                          # models/resnet_passport.py:32-65
        """Each passport layer records what feeds it in the plain block; order convbnrelu_1, convbn_2, shortcut."""
        if isinstance(self.convbnrelu_1, PassportLayerRef):
            self.convbnrelu_1.set_key(x, y)
        ox, oy = plain_block.convbnrelu_1(x), plain_block.convbnrelu_1(y)
        if isinstance(self.convbn_2, PassportLayerRef):
            self.convbn_2.set_key(ox, oy)
        ox, oy = plain_block.convbn_2(ox), plain_block.convbn_2(oy)
        if not isinstance(self.shortcut, nn.Sequential):
            if isinstance(self.shortcut, PassportLayerRef):
                self.shortcut.set_key(x, y)
            ox, oy = ox + plain_block.shortcut(x), oy + plain_block.shortcut(y)
        else:
            ox, oy = ox + x, oy + y
        return F.relu(ox), F.relu(oy)

    def forward(self, x, force_passport=False, ind=0):
        out = _call(self.convbnrelu_1, x, force_passport, ind)
        out = _call(self.convbn_2, out, force_passport, ind)
        if not isinstance(self.shortcut, nn.Sequential):
            out = out + _call(self.shortcut, x, force_passport, ind)
        else:
            out = out + x
        return F.relu(out)


class BottleneckRef(nn.Module):
    """NOT in the reference: a passport Bottleneck composed from the reference's pieces the way its plain
    Bottleneck is (models/resnet_normal.py:30-49): 1x1 -> 3x3 -> 1x1 (no ReLU) + projection (no ReLU), ReLU after
    the add.  Checker for the build's BottleneckPassportBlock (BASELINE.json config 5 has no reference oracle)."""
    expansion = 4

    def __init__(self, in_planes, planes, stride, kw, private):
        super().__init__()

        def make(k, i, o, ks, s, pd, relu):
            if k['flag']:
                return PassportLayerRef(i, o, ks, s, pd, passport_kwargs=k, relu=relu, private=private)
            return ConvBlockRef(i, o, ks, s, pd, bn=k['norm_type'], relu=relu)
        self.convbnrelu_1 = make(kw['convbnrelu_1'], in_planes, planes, 1, 1, 0, True)
        self.convbnrelu_2 = make(kw['convbnrelu_2'], planes, planes, 3, stride, 1, True)
        self.convbn_3 = make(kw['convbn_3'], planes, 4 * planes, 1, 1, 0, False)
        self.shortcut = nn.Sequential()
        if stride != 1 or in_planes != 4 * planes:
            self.shortcut = make(kw['shortcut'], in_planes, 4 * planes, 1, stride, 0, False)

    def forward(self, x, force_passport=False, ind=0):
        out = _call(self.convbnrelu_1, x, force_passport, ind)
        out = _call(self.convbnrelu_2, out, force_passport, ind)
        out = _call(self.convbn_3, out, force_passport, ind)
        sc = x if isinstance(self.shortcut, nn.Sequential) else _call(self.shortcut, x, force_passport, ind)
        return F.relu(out + sc)


class ResNetRef(nn.Module):
    """ResNetPassport (models/resnet_passport.py:88-180) / ResNetPrivate
    (models/resnet_passport_private.py:89-182)."""

    def __init__(self, num_blocks, num_classes=10, passport_kwargs={}, private=False, imagenet=False,
                 block=None):
        super().__init__()
        block = block or BasicBlockRef
        exp = getattr(block, 'expansion', 1)
        self.in_planes = 64
        stem = _make_block(passport_kwargs['convbnrelu_1'], private)
        if num_classes == 1000 or imagenet:                                       # :94-98
            self.convbnrelu_1 = nn.Sequential(stem(3, 64, 7, 2, 3), nn.MaxPool2d(3, 2, 1))
        else:
            self.convbnrelu_1 = stem(3, 64, 3, 1, 1)
        for li, (planes, stride) in enumerate([(64, 1), (128, 2), (256, 2), (512, 2)]):
            name = 'layer%d' % (li + 1)
            blocks = []
            for bi, s in enumerate([stride] + [1] * (num_blocks[li] - 1)):         # :137-143
                blocks.append(block(self.in_planes, planes, s, passport_kwargs[name][str(bi)], private))
                self.in_planes = planes * exp
            setattr(self, name, nn.Sequential(*blocks))
        self.linear = nn.Linear(512 * exp, num_classes)

    def set_intermediate_keys(self, plain, x, y):                                # :145-161
        with torch.no_grad():
            if isinstance(self.convbnrelu_1, PassportLayerRef):                  # (a Sequential stem is skipped, :147)
                self.convbnrelu_1.set_key(x, y)
            x, y = plain.convbnrelu_1(x), plain.convbnrelu_1(y)
            for name in ('layer1', 'layer2', 'layer3', 'layer4'):
                for mine, theirs in zip(getattr(self, name), getattr(plain, name)):
                    x, y = mine.set_intermediate_keys(theirs, x, y)

    def forward(self, x, force_passport=False, ind=0):                           # :163-180
        if isinstance(self.convbnrelu_1, nn.Sequential):
            out = self.convbnrelu_1[1](_call(self.convbnrelu_1[0], x, force_passport, ind))
        else:
            out = _call(self.convbnrelu_1, x, force_passport, ind)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                out = blk(out, force_passport, ind)
        out = F.adaptive_avg_pool2d(out, (1, 1)).view(out.size(0), -1)
        return self.linear(out)


def resnet18_ref(**kw):
    return ResNetRef([2, 2, 2, 2], **kw)


def resnet9_ref(**kw):
    return ResNetRef([1, 1, 1, 1], **kw)


def resnet50_ref(**kw):
    return ResNetRef([3, 4, 6, 3], block=BottleneckRef, **kw)


# ----------------------------------------------------------------------------- AlexNet
class AlexNetRef(nn.Module):
    """AlexNetPassport (models/alexnet_passport.py:9-122) / AlexNetPassportPrivate
    (models/alexnet_passport_private.py:9-121), CIFAR geometry (num_classes != 1000)."""

    def __init__(self, in_channels, num_classes, passport_kwargs, private=False):
        super().__init__()
        oups = {0: 64, 2: 192, 4: 384, 5: 256, 6: 256}                            # :22-28
        kp = {0: (5, 2), 2: (5, 2), 4: (3, 1), 5: (3, 1), 6: (3, 1)}              # :29-35
        layers, inp = [], in_channels
        for idx in range(8):
            if idx in (1, 3, 7):                                                  # :16,38-40
                layers.append(nn.MaxPool2d(2, 2))
                continue
            k, p = kp[idx]
            kw = passport_kwargs[str(idx)]
            if kw['flag']:
                layers.append(PassportLayerRef(inp, oups[idx], k, 1, p, kw, private=private))
            else:
                layers.append(ConvBlockRef(inp, oups[idx], k, 1, p, kw['norm_type']))
            inp = oups[idx]
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Linear(4 * 4 * 256, num_classes)                     # :69

    def set_intermediate_keys(self, plain, x, y):                                # :104-112
        with torch.no_grad():
            for theirs, mine in zip(plain.features, self.features):
                if isinstance(mine, PassportLayerRef):
                    mine.set_key(x, y)
                x, y = theirs(x), theirs(y)

    def forward(self, x, force_passport=False, ind=0):                           # :114-122
        for m in self.features:
            x = _call(m, x, force_passport, ind)
        return self.classifier(x.view(x.size(0), -1))


def plain_net(arch, num_classes, norm_type='bn'):
    """The un-passported twin used to propagate keys (models/resnet_normal.py:9-27,52-71,126-127;
    models/alexnet_normal.py:52-64): the same nets with every flag off -- identical module names."""
    off = {'flag': False, 'norm_type': norm_type, 'key_type': 'random', 'sign_loss': 0}
    if arch == 'alexnet':
        return AlexNetRef(3, num_classes, {str(i): off for i in (0, 2, 4, 5, 6)})
    blk = {'convbnrelu_1': off, 'convbn_2': off, 'shortcut': off}
    kw = {'convbnrelu_1': off}
    for li in (1, 2, 3, 4):
        kw['layer%d' % li] = {'0': blk, '1': blk}
    if arch == 'resnet9':
        return resnet9_ref(num_classes=num_classes, passport_kwargs=kw)
    return resnet18_ref(num_classes=num_classes, passport_kwargs=kw)


# ----------------------------------------------------------------------------- config -> kwargs
def passport_kwargs_from_config(cfg, norm_type='bn', key_type='random', sl_ratio=0.1):
    """experiments/utils.py:6-50: JSON tree -> {'flag','norm_type','key_type','sign_loss'[, 'b']}."""
    def leaf(v):
        d = {'flag': True if isinstance(v, str) else v, 'norm_type': norm_type,
             'key_type': key_type, 'sign_loss': sl_ratio}
        if isinstance(v, str):
            d['b'] = v
        return d

    def walk(node):
        if isinstance(node, dict):
            return {k: walk(v) for k, v in node.items()}
        return leaf(node)
    return walk(cfg)


# ----------------------------------------------------------------------------- train steps
def sign_losses(model):
    return [m for m in model.modules() if isinstance(m, SignLossRef)]


def v1_step(model, optimizer, data, target, wm=None):
    """One batch of Trainer.train: experiments/trainer.py:111-149."""
    if wm is not None:                                                           # :115-126
        data = torch.cat([data, wm[0]], dim=0)
        target = torch.cat([target, wm[1]], dim=0)
    optimizer.zero_grad()                                                        # :128
    for m in sign_losses(model):                                                 # :131-133
        m.reset()
    pred = model(data)                                                           # :135
    loss = F.cross_entropy(pred, target)                                         # :136
    sign_loss = torch.tensor(0.)
    for m in sign_losses(model):                                                 # :140-142
        sign_loss = sign_loss + m.loss
    (loss + sign_loss).backward()                                                # :144
    optimizer.step()                                                             # :145
    return {'pred': pred.detach(), 'loss': loss.detach(), 'sign_loss': sign_loss.detach()}


def v23_step(model, optimizer, data, target, wm=None):
    """One batch of TrainerPrivate.train: experiments/trainer_private.py:131-177 (two forwards, one backward)."""
    if wm is not None:                                                           # :135-146
        data = torch.cat([data, wm[0]], dim=0)
        target = torch.cat([target, wm[1]], dim=0)
    optimizer.zero_grad()
    for m in sign_losses(model):
        m.reset()
    loss = torch.tensor(0.)
    preds = []
    for ind in range(2):                                                         # :159-166
        pred = model(data, ind=ind)
        loss = loss + F.cross_entropy(pred, target)
        preds.append(pred.detach())
    sign_loss = torch.tensor(0.)
    for m in sign_losses(model):                                                 # :169-171
        sign_loss = sign_loss + m.loss
    (loss + sign_loss).backward()                                                # :173
    optimizer.step()
    return {'pred_public': preds[0], 'pred_private': preds[1], 'loss': loss.detach(),
            'sign_loss': sign_loss.detach()}


def signature_report(model):
    """TesterPrivate.test_signature: experiments/trainer_private.py:37-71.  Returns
    {name: (sign(gamma) int8 [C], detection rate)} with the private / public prefix of the reference."""
    out = {}
    model.eval()
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, PassportLayerRef):
                g = (m.get_scale(ind=1) if m.private else m.get_scale()).view(-1)
                bits = g.sign()
                key = ('private_' if m.private else 'public_') + name
                out[key] = (bits.to(torch.int8), (bits == m.b).float().mean().item())
    return out
