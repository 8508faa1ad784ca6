"""Directory / tensor helpers -- same API as the reference's utils/misc.py:10-34."""
import os

import numpy as np
import torch


def mkdir(path):
    if not os.path.exists(path):
        os.makedirs(path, exist_ok=True)


def mkdirs(*paths):
    for path in paths:
        mkdir(path)


def to_numpy(input):
    if isinstance(input, torch.Tensor):
        return input.detach().cpu().numpy()
    if isinstance(input, np.ndarray):
        return input
    raise TypeError('Unknown type of input, expected torch.Tensor or np.ndarray, but got {}'.format(type(input)))


def module_size(module):
    assert isinstance(module, torch.nn.Module)
    n_params, n_conv_layers = 0, 0
    for name, param in module.named_parameters():
        if 'conv' in name:
            n_conv_layers += 1
        n_params += param.numel()
    return n_params, n_conv_layers
