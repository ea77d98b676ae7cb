"""Passport image sampling and key installation (reference passport_generator.py:6-43).

One-time setup, not on the per-step path: n images are drawn from a loader's dataset and pushed through a
(pre-trained) plain network; every passport layer receives the activations that feed it
(`set_intermediate_keys`), reduced to one [1,C,H,W] passport by `passport_selection`."""
import random

import torch


def get_key(dataset_loader, n=32):
    """n distinct random samples of the loader's dataset, stacked -> ([n,C,H,W], indices)."""
    dataset = dataset_loader.dataset
    indices = random.sample(range(len(dataset)), n)
    return torch.cat([dataset[i][0].unsqueeze(0) for i in indices], dim=0), indices


def get_intermediate_key(input_key, intermediate_key_name, pretrained_model):
    """Activation entering `features.<i>` of a plain AlexNet (passport_generator.py:20-27)."""
    x = input_key
    with torch.no_grad():
        for i, m in enumerate(pretrained_model.features):
            if 'features.%d' % i == intermediate_key_name:
                return x
            x = m(x)


def set_key(pretrained_model, target_model, key_x, key_y, ind=None):
    """key_x -> bias keys, key_y -> scale keys of every passport layer of target_model."""
    if key_x.dim() == 3:
        key_x = key_x.unsqueeze(0)
    if key_y is not None and key_y.dim() == 3:
        key_y = key_y.unsqueeze(0)
    if ind is not None:
        target_model.set_intermediate_keys(pretrained_model, key_x, key_y, ind)
    else:
        target_model.set_intermediate_keys(pretrained_model, key_x, key_y)
