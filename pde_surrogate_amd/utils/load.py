"""Dataset loading -- same contract as the reference's utils/load.py:18-37 (HDF5 with datasets
`input` (n,1,H,W) and `output` (n,3,H,W) -> fp32 TensorDataset -> DataLoader(shuffle, drop_last)),
plus the device-resident loader the fused trainer uses and a synthetic source for machines without
the (non-redistributed) datasets.

`h5py` is optional: it is only imported when an .hdf5 file is actually opened, and when it is not installed the
files are read by utils/hdf5_lite.py (pure Python: superblock v0-v3, contiguous / chunked + deflate datasets);
`.npz` files with the same two arrays are accepted too."""
import json
import os
from argparse import Namespace

import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset


def load_args(run_dir):
    with open(run_dir + '/args.txt') as args_file:
        return Namespace(**json.load(args_file))


def read_arrays(path, ndata, only_input=True):
    """(x, y or None) from an .hdf5 / .npz file holding `input` and `output`.  When the .hdf5 file named by the
    reference's path convention is absent but an .npz with the same stem exists, that one is read."""
    if not os.path.exists(path) and os.path.exists(os.path.splitext(path)[0] + '.npz'):
        path = os.path.splitext(path)[0] + '.npz'
    if path.endswith('.npz'):
        with np.load(path) as f:
            x = f['input'][:ndata]
            y = None if only_input else f['output'][:ndata]
        return x, y
    try:
        import h5py
    except ImportError:
        from . import hdf5_lite as h5py          # the build's own reader of the subset of HDF5 the datasets use
    with h5py.File(path, 'r') as f:
        x = f['input'][:ndata]
        y = None if only_input else f['output'][:ndata]
    return x, y


def y_variation(y):
    """sum over (n,h,w) of (y - mean_n y)^2 per channel (load.py:28-30), used for the R^2 score"""
    y = np.asarray(y)
    return ((y - y.mean(0, keepdims=True)) ** 2).sum(axis=(0, 2, 3))


def load_data(hdf5_file, ndata, batch_size, only_input=True, return_stats=False):
    """reference signature: returns (DataLoader, stats)"""
    x_data, y_data = read_arrays(hdf5_file, ndata, only_input)
    print(f'x_data: {x_data.shape}')
    if not only_input:
        print(f'y_data: {y_data.shape}')
    stats = {}
    if return_stats:
        stats['y_variation'] = y_variation(y_data)
    data_tuple = (torch.FloatTensor(x_data),) if only_input else (torch.FloatTensor(x_data), torch.FloatTensor(y_data))
    data_loader = DataLoader(TensorDataset(*data_tuple), batch_size=batch_size, shuffle=True, drop_last=True)
    print(f'Loaded dataset: {hdf5_file}')
    return data_loader, stats


def reference_epoch_order(n):
    """the permutation of `n` samples the reference's `DataLoader(TensorDataset(...), shuffle=True)` (utils/load.py:34-35)
    uses for its NEXT epoch, drawn exactly as torch draws it -- through a real DataLoader over the indices, so whatever the
    installed torch takes from the GLOBAL generator when an iterator is created (the iterator's base seed, the
    RandomSampler's seed) is taken here too, in the same order.  With the reference's seed and creation order (Parser:
    manual_seed -> DenseED -> loaders) a run of this build therefore sees the reference's minibatches, epoch by epoch
    (tests/golden/G25: the permutations of the reference's own run)."""
    loader = DataLoader(range(n), batch_size=n, shuffle=True, collate_fn=lambda b: b)
    return torch.tensor(next(iter(loader)), dtype=torch.int64)


class DeviceLoader:
    """Device-resident replacement for DataLoader(shuffle=True, drop_last=True): the whole dataset
    lives in HBM (4096 x 16 KiB = 64 MiB), a minibatch is one index_select.  With world_size > 1 every
    rank holds the replica and takes its contiguous slice of each global batch from a permutation
    that is identical on all ranks (same generator seed).  Like the reference's loader (utils/load.py:34-35 there:
    drop_last=True) samples beyond the last full global batch of an epoch's permutation are dropped."""

    def __init__(self, *tensors, batch_size, device, shuffle=True, seed=0, rank=0, world_size=1, order='own'):
        self.tensors = [t.to(device) for t in tensors]
        self.n = self.tensors[0].shape[0]
        self.batch_size, self.rank, self.world = batch_size, rank, world_size
        self.global_batch = batch_size * world_size
        if self.n < self.global_batch:
            raise ValueError(f'{self.n} samples are fewer than one global batch of {self.global_batch}')
        self.shuffle = shuffle
        # order='reference': every epoch's permutation comes from torch's global generator the way the reference's
        # DataLoader draws it (reference_epoch_order); 'own': from a generator of this loader seeded with `seed`
        if order not in ('own', 'reference'):
            raise ValueError("DeviceLoader order: 'own' or 'reference'")
        self.order = order
        self.gen = torch.Generator(device='cpu').manual_seed(seed)
        self.device = device

    def __len__(self):
        return self.n // self.global_batch

    def __iter__(self):
        if not self.shuffle:
            perm = torch.arange(self.n)
        elif self.order == 'reference':
            perm = reference_epoch_order(self.n)
        else:
            perm = torch.randperm(self.n, generator=self.gen)
        perm = perm.to(self.device)
        for i in range(len(self)):
            lo = i * self.global_batch + self.rank * self.batch_size
            idx = perm[lo:lo + self.batch_size]
            yield tuple(t.index_select(0, idx) for t in self.tensors)
