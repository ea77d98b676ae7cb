"""Implementation adapters for oracle.runner.collect (see its docstring)."""
import numpy as np
import torch

from oracle import torch_ref
from oracle.cases import ALPHA


def load_golden(golden_dir, name):
    import os
    z = np.load(os.path.join(golden_dir, name + '.npz'))
    return {k.replace('|', '/'): z[k] for k in z.files}


class OracleImpl:
    """oracle/torch_ref.py on the host CPU."""
    device = torch.device('cpu')

    def build(self, case):
        kw = torch_ref.passport_kwargs_from_config(case['config'], case['norm'], case.get('key_type', 'random'),
                                                   ALPHA)
        private = case['scheme'] != 1
        if case['arch'] == 'alexnet':
            return torch_ref.AlexNetRef(3, case['ncls'], kw, private=private)
        if case['arch'] == 'resnet9':
            return torch_ref.resnet9_ref(num_classes=case['ncls'], passport_kwargs=kw, private=private)
        return torch_ref.resnet18_ref(num_classes=case['ncls'], passport_kwargs=kw, private=private)

    def plain(self, case):
        return torch_ref.plain_net(case['arch'], case['ncls'], case['norm'])

    def set_keys(self, plain, model, kx, ky):
        model.set_intermediate_keys(plain, kx, ky)

    def is_passport(self, m):
        return isinstance(m, torch_ref.PassportLayerRef)

    def is_private(self, m):
        return isinstance(m, torch_ref.PassportLayerRef) and m.private

    def step(self, model, opt, batch, wm):
        private = any(self.is_private(m) for m in model.modules())
        model.train()
        r = (torch_ref.v23_step if private else torch_ref.v1_step)(model, opt, batch[0], batch[1], wm)
        sl = torch_ref.sign_losses(model)
        acc = sum(float(m.acc) for m in sl) / max(1, len(sl))
        return {'loss': float(r['loss']), 'sign_loss': float(r['sign_loss']), 'sign_acc': acc}

    def test_signature(self, model):
        return {k: v[1] for k, v in torch_ref.signature_report(model).items()}


class ProductImpl:
    """The HIP build (deepipr_amd).  On the GPU box it runs the real kernels on cuda:0; the CPU suite
    passes device='cpu' after monkeypatching passport_ops.kernels with tests/oracle_kernels.py."""

    def __init__(self, device='cuda:0', fuse_norm=True):
        self.device = torch.device(device)
        self.fuse_norm = fuse_norm

    def build(self, case):
        from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
        from deepipr_amd.models.alexnet_passport import AlexNetPassport
        from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
        from deepipr_amd.models.resnet_passport import ResNet9Passport, ResNet18Passport
        from deepipr_amd.models.resnet_passport_private import ResNet18Private
        kw = construct_passport_kwargs_from_dict({'passport_config': case['config'], 'norm_type': case['norm'],
                                                  'key_type': case.get('key_type', 'random'), 'sl_ratio': ALPHA})
        private = case['scheme'] != 1
        if case['arch'] == 'alexnet':
            model = (AlexNetPassportPrivate if private else AlexNetPassport)(3, case['ncls'], kw)
        elif case['arch'] == 'resnet9':
            model = ResNet9Passport(num_classes=case['ncls'], passport_kwargs=kw)
        else:
            model = (ResNet18Private if private else ResNet18Passport)(num_classes=case['ncls'], passport_kwargs=kw)
        for m in model.modules():
            if self.is_passport(m):
                m.fuse_norm = self.fuse_norm     # BatchNorm folded into the passport kernels, or the unfused ops
        return model.to(self.device)

    def plain(self, case):
        from deepipr_amd.models.alexnet_normal import AlexNetNormal
        from deepipr_amd.models.resnet_normal import ResNet9, ResNet18
        if case['arch'] == 'alexnet':
            return AlexNetNormal(3, case['ncls'], case['norm']).to(self.device)
        if case['arch'] == 'resnet9':
            return ResNet9(num_classes=case['ncls'], norm_type=case['norm']).to(self.device)
        return ResNet18(num_classes=case['ncls'], norm_type=case['norm']).to(self.device)

    def set_keys(self, plain, model, kx, ky):
        from deepipr_amd import passport_generator
        passport_generator.set_key(plain, model, kx, ky)

    def is_passport(self, m):
        from deepipr_amd.models._builders import PASSPORT_TYPES
        return isinstance(m, PASSPORT_TYPES)

    def is_private(self, m):
        from deepipr_amd.models.layers.passportconv2d_private import PassportPrivateBlock
        return isinstance(m, PassportPrivateBlock)

    def step(self, model, opt, batch, wm):
        from deepipr_amd.experiments.trainer import Trainer
        from deepipr_amd.experiments.trainer_private import TrainerPrivate
        private = any(self.is_private(m) for m in model.modules())
        tr = (TrainerPrivate if private else Trainer)(model, opt, None, self.device)
        return tr.train(0, [batch], [wm] if wm is not None else None)

    def test_signature(self, model):
        from deepipr_amd.experiments.trainer_private import TesterPrivate
        return TesterPrivate(model, self.device, verbose=False).test_signature()


class ProductShuttle:
    """deepipr_amd.experiments.utils weight shuttles on deepipr_amd's nets (oracle.runner.collect_shuttle)."""

    def __init__(self, device='cpu'):
        self.device = torch.device(device)

    def _kw(self, arch):
        from deepipr_amd.experiments.utils import construct_passport_kwargs_from_dict
        from oracle.cases import alexnet_config, resnet18_config
        cfg = alexnet_config() if arch == 'alexnet' else resnet18_config()
        return construct_passport_kwargs_from_dict({'passport_config': cfg, 'norm_type': 'bn',
                                                    'key_type': 'random', 'sl_ratio': ALPHA}, True)

    def plkeys(self, arch):
        return self._kw(arch)[1]

    def plain(self, arch, ncls):
        from deepipr_amd.models.alexnet_normal import AlexNetNormal
        from deepipr_amd.models.resnet_normal import ResNet18
        m = AlexNetNormal(3, ncls, 'bn') if arch == 'alexnet' else ResNet18(num_classes=ncls, norm_type='bn')
        return m.to(self.device)

    def passport(self, arch, ncls, private):
        from deepipr_amd.models.alexnet_passport import AlexNetPassport
        from deepipr_amd.models.alexnet_passport_private import AlexNetPassportPrivate
        from deepipr_amd.models.resnet_passport import ResNet18Passport
        from deepipr_amd.models.resnet_passport_private import ResNet18Private
        kw = self._kw(arch)[0]
        if arch == 'alexnet':
            m = (AlexNetPassportPrivate if private else AlexNetPassport)(3, ncls, kw)
        else:
            m = (ResNet18Private if private else ResNet18Passport)(num_classes=ncls, passport_kwargs=kw)
        return m.to(self.device)

    def n2p(self, *a):
        from deepipr_amd.experiments.utils import load_normal_model_to_passport_model
        return load_normal_model_to_passport_model(*a)

    def n2n(self, *a):
        from deepipr_amd.experiments.utils import load_normal_model_to_normal_model
        return load_normal_model_to_normal_model(*a)

    def p2n(self, *a):
        from deepipr_amd.experiments.utils import load_passport_model_to_normal_model
        return load_passport_model_to_normal_model(*a)
