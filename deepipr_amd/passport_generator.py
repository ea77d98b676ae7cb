"""Passport sampling and key installation (the one-time set-up behind `--key-type image|shuffle`).

Behavioural twin of the reference's passport_generator.py:6-43, organised around one small class:
a `PassportSampler` draws candidate images from a loader's dataset with python's `random` (so seeded runs pick
the same images as the reference) and `install()` pushes them through a plain "key propagation" network while
every passport layer of the target network records the activations that feed it (`set_intermediate_keys`,
which reduces 20 candidates to one passport per layer through `passport_selection`).
"""
import random

import torch


class PassportSampler:
    def __init__(self, dataset_loader):
        self.dataset = dataset_loader.dataset

    def draw(self, n):
        """n distinct samples -> (stacked images [n,C,H,W], their dataset indices)."""
        picked = random.sample(range(len(self.dataset)), n)
        images = torch.stack([self.dataset[idx][0] for idx in picked], dim=0)
        return images, picked

    @staticmethod
    def install(key_net, target_net, bias_keys, scale_keys=None, ind=None):
        """bias_keys feed `key` (beta), scale_keys feed `skey` (gamma) of every passport layer of target_net."""
        bias_keys = _as_batch(bias_keys)
        scale_keys = None if scale_keys is None else _as_batch(scale_keys)
        extra = () if ind is None else (ind,)
        target_net.set_intermediate_keys(key_net, bias_keys, scale_keys, *extra)


def _as_batch(t):
    return t.unsqueeze(0) if t.dim() == 3 else t


# ---- the reference's function-style entry points ---------------------------------------------------------
def get_key(dataset_loader, n=32):
    return PassportSampler(dataset_loader).draw(n)


def set_key(pretrained_model, target_model, key_x, key_y, ind=None):
    PassportSampler.install(pretrained_model, target_model, key_x, key_y, ind)


def get_intermediate_key(input_key, intermediate_key_name, pretrained_model):
    """Activation that enters `features.<i>` of a plain AlexNet-style net, or None if no layer has that name."""
    act = input_key
    with torch.no_grad():
        for idx, layer in enumerate(pretrained_model.features):
            if intermediate_key_name == 'features.%d' % idx:
                return act
            act = layer(act)
    return None
