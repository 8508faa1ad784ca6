"""CPU tests of the dataset wire format (reference utils/load.py:18-37): the HDF5 branch of
`read_arrays` / `load_data` executes here WITHOUT h5py, through pde_surrogate_amd/utils/hdf5_lite.py.

The reader is checked against (a) a file written by the HDF5 library itself that ships with scipy -- the MATLAB 7.3
test file `testhdf5_7.4_GLNX86.mat` (512-byte user block, superblock v0, symbol-table group, contiguous float64
dataset), whose content scipy reads from the sibling MATLAB-5 file -- and (b) round trips through the module's own
minimal writer (contiguous and chunked + deflate layouts)."""
import os

import numpy as np
import pytest
import torch

from pde_surrogate_amd.utils import hdf5_lite
from pde_surrogate_amd.utils.load import load_data, read_arrays, y_variation


def _scipy_data(name):
    import scipy.io
    return os.path.join(os.path.dirname(scipy.io.__file__), 'matlab', 'tests', 'data', name)


def test_reads_a_file_written_by_the_hdf5_library():
    import scipy.io
    path = _scipy_data('testhdf5_7.4_GLNX86.mat')
    if not os.path.exists(path):
        pytest.skip('scipy test data not installed')
    with hdf5_lite.File(path, 'r') as f:
        assert f.keys() == ['testdouble']
        d = f['testdouble']
        assert d.shape == (9, 1) and d.dtype == np.float64
        got = d[()]
    want = scipy.io.loadmat(_scipy_data('testdouble_7.4_GLNX86.mat'))['testdouble']       # (1, 9) in MATLAB order
    np.testing.assert_array_equal(got.T, want)
    np.testing.assert_allclose(got[:, 0], np.linspace(0, 2 * np.pi, 9), rtol=1e-15)


@pytest.mark.parametrize('layout', ['contiguous', 'chunked-gzip'])
def test_round_trip_of_the_reference_dataset_layout(tmp_path, layout):
    rng = np.random.default_rng(3)
    x = np.exp(0.5 * rng.standard_normal((20, 1, 16, 16))).astype(np.float32)
    y = rng.standard_normal((20, 3, 16, 16))                       # float64 on disk, cast to fp32 by the loader
    idx = np.arange(7, dtype=np.int64)
    path = str(tmp_path / 'kle512_lhs10000_train.hdf5')
    kw = {}
    if layout != 'contiguous':
        kw = dict(chunks={'input': (8, 1, 16, 16), 'output': (6, 3, 16, 8)}, compression='gzip')   # ragged edge chunks
    hdf5_lite.write_hdf5(path, {'input': x, 'output': y, 'idx': idx}, **kw)
    with hdf5_lite.File(path) as f:
        assert sorted(f.keys()) == ['idx', 'input', 'output']
        assert f['input'].shape == (20, 1, 16, 16) and f['input'].dtype == np.float32
        np.testing.assert_array_equal(f['input'][:5], x[:5])
        np.testing.assert_array_equal(f['output'][()], y)
        np.testing.assert_array_equal(f['idx'][()], idx)
        with pytest.raises(KeyError):
            f['nope']
    # the reference's loader contract (utils/load.py:18-37) through the HDF5 branch
    xs, ys = read_arrays(path, 12, only_input=False)
    np.testing.assert_array_equal(xs, x[:12])
    np.testing.assert_array_equal(ys, y[:12])
    loader, stats = load_data(path, 12, 4, only_input=False, return_stats=True)
    np.testing.assert_allclose(stats['y_variation'], y_variation(y[:12]), rtol=1e-12)
    batches = list(loader)
    assert len(batches) == 3 and batches[0][0].shape == (4, 1, 16, 16) and batches[0][1].dtype == torch.float32
    xs2, none = read_arrays(path, 3, only_input=True)
    assert none is None and xs2.shape == (3, 1, 16, 16)


def test_rejects_what_it_does_not_implement(tmp_path):
    p = tmp_path / 'not.hdf5'
    p.write_bytes(b'hello world' * 100)
    with pytest.raises(OSError, match='not an HDF5 file'):
        hdf5_lite.File(str(p))
    with pytest.raises(ValueError):
        hdf5_lite.File(str(p), 'w')


# ---- files written by the HDF5 library itself (h5py 3.3 / libhdf5 1.10.6; tools/gen_hdf5_fixtures.py) ------------------
FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'hdf5')
FOREIGN = ['default_contiguous.hdf5', 'chunked_gzip_shuffle.hdf5', 'btree_depth2.hdf5', 'latest_fixed_array.hdf5',
           'latest_fixed_array_paged.hdf5', 'bigendian_userblock.hdf5', 'resizable_earliest.hdf5', 'resizable_latest.hdf5']
REFUSED = {('resizable_latest.hdf5', 'input'): 'extensible array'}


@pytest.mark.parametrize('name', FOREIGN)
def test_reads_library_written_layouts(name):
    """every layout h5py emits for `create_dataset(name, data=..., [chunks, compression, shuffle, fletcher32])` with the
    default and with libver='latest' (the reader is checked against bytes it did not write);
    what it does not implement it must refuse BY NAME"""
    import hashlib
    import json
    path = os.path.join(FIX, name)
    manifest = json.load(open(os.path.join(FIX, 'MANIFEST.json')))
    assert hashlib.sha256(open(path, 'rb').read()).hexdigest() == manifest[name]['sha256']     # the library's bytes, untouched
    exp = np.load(path + '.expected.npz')
    with hdf5_lite.File(path) as f:
        for key in exp.files:
            why = REFUSED.get((name, key))
            if why:
                with pytest.raises(NotImplementedError, match=why):
                    f[key][...]
                continue
            d = f[key]
            assert d.shape == exp[key].shape and d.dtype.itemsize == exp[key].dtype.itemsize
            np.testing.assert_array_equal(np.asarray(d[...], exp[key].dtype), exp[key])
            np.testing.assert_array_equal(np.asarray(d[:3], exp[key].dtype), exp[key][:3])


def test_loader_contract_on_a_library_written_file():
    """utils/load.py:18-37 of the reference (`f['input'][:ndata]`, fp32 tensors, y_variation) through a file h5py wrote"""
    path = os.path.join(FIX, 'chunked_gzip_shuffle.hdf5')
    exp = np.load(path + '.expected.npz')
    x, y = read_arrays(path, 10, only_input=False)
    np.testing.assert_array_equal(x, exp['input'][:10])
    np.testing.assert_array_equal(y, exp['output'][:10])
    loader, stats = load_data(path, 8, 4, only_input=False, return_stats=True)
    np.testing.assert_allclose(stats['y_variation'], y_variation(exp['output'][:8]), rtol=1e-6)
    assert len(loader) == 2 and next(iter(loader))[0].dtype == torch.float32
