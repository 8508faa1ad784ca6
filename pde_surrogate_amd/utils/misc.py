"""Host helpers with the call surface of the reference's utils/misc.py (mkdirs :10-18, to_numpy :21-28,
module_size :31-38), written for this package: pathlib for the run directories, one pass over
`named_parameters()` for the (parameter count, conv-layer count) pair that goes into args.txt."""
from pathlib import Path

import numpy as np
import torch


def mkdirs(*paths):
    """create every run directory (parents included); existing ones are left alone"""
    for p in paths:
        Path(p).mkdir(parents=True, exist_ok=True)


mkdir = mkdirs


def to_numpy(value):
    """device or host tensor / ndarray -> ndarray on the host (no copy for an ndarray)"""
    if torch.is_tensor(value):
        return value.detach().to('cpu').numpy()
    if not isinstance(value, np.ndarray):
        raise TypeError(f'to_numpy expects a torch.Tensor or np.ndarray, got {type(value).__name__}')
    return value


def module_size(module):
    """(number of scalar parameters, number of parameter tensors whose qualified name contains 'conv');
    DenseED keeps the reference's layer names so the second figure is 28 for the default net."""
    if not isinstance(module, torch.nn.Module):
        raise TypeError('module_size expects an nn.Module')
    sizes = {name: p.numel() for name, p in module.named_parameters()}
    return sum(sizes.values()), sum('conv' in name for name in sizes)
