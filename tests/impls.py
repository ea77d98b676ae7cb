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
        kw = torch_ref.passport_kwargs_from_config(case['config'], case['norm'], 'random', ALPHA)
        private = case['scheme'] != 1
        if case['arch'] == 'alexnet':
            return torch_ref.AlexNetRef(3, case['ncls'], kw, private=private)
        return torch_ref.resnet18_ref(num_classes=case['ncls'], passport_kwargs=kw, private=private)

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
