#!/opt/conda/bin/python3.9
"""Write the HDF5 fixtures of tests/golden/hdf5/ with the REAL HDF5 library (h5py 3.3.0 / libhdf5 1.10.6 of this
container's /opt/conda -- the product's interpreter has no h5py): the layouts h5py emits for the reference's dataset
files (utils/load.py:18-37 reads `f['input'][:n]`, `f['output'][:n]`).  Each file carries its expected content in a
sibling .npz (arrays re-created from the same seed here), so tests/test_hdf5_lite_cpu.py checks the pure-Python
reader against bytes it did not write.

    /opt/conda/bin/python3.9 tools/gen_hdf5_fixtures.py
"""
import hashlib
import json
import os

import h5py
import numpy as np

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'hdf5')
os.makedirs(OUT, exist_ok=True)
rng = np.random.default_rng(20190701)
manifest = {}


def arrays(n, hw, out_dtype='f8'):
    x = np.exp(0.5 * rng.standard_normal((n, 1, hw, hw))).astype('f4')
    y = rng.standard_normal((n, 3, hw, hw)).astype(out_dtype)
    return x, y


def done(name, content):
    path = os.path.join(OUT, name)
    np.savez_compressed(path + '.expected.npz', **content)
    manifest[name] = {'sha256': hashlib.sha256(open(path, 'rb').read()).hexdigest(), 'bytes': os.path.getsize(path),
                      'writer': f'h5py {h5py.__version__} / HDF5 {h5py.version.hdf5_version}'}


# 1. what `h5py.File(p, 'w').create_dataset(name, data=a)` writes: superblock v0, symbol-table group, contiguous
x, y = arrays(12, 8)
with h5py.File(os.path.join(OUT, 'default_contiguous.hdf5'), 'w') as f:
    f.create_dataset('input', data=x)
    d = f.create_dataset('output', data=y)
    d.attrs['units'] = 'SI'                      # attribute messages must be skipped
    f.attrs['generator'] = 'fenics'
done('default_contiguous.hdf5', {'input': x, 'output': y})

# 2. chunked + gzip + shuffle, B-tree v1 chunk index (libver earliest); edge chunks (12 is not a multiple of 5)
x, y = arrays(12, 8, 'f4')
with h5py.File(os.path.join(OUT, 'chunked_gzip_shuffle.hdf5'), 'w') as f:
    f.create_dataset('input', data=x, chunks=(5, 1, 8, 8), compression='gzip', compression_opts=4, shuffle=True)
    f.create_dataset('output', data=y, chunks=(5, 2, 4, 8), compression='gzip', shuffle=False, fletcher32=True)
done('chunked_gzip_shuffle.hdf5', {'input': x, 'output': y})

# 3. many chunks: a B-tree v1 of depth 2 (more than 64 entries per node)
x = rng.standard_normal((300, 1, 2, 2)).astype('f4')
with h5py.File(os.path.join(OUT, 'btree_depth2.hdf5'), 'w') as f:
    f.create_dataset('input', data=x, chunks=(1, 1, 2, 2), compression='gzip')
done('btree_depth2.hdf5', {'input': x})

# 4. libver='latest': superblock v3, object headers v2, link messages, v4 layout with a FIXED ARRAY chunk index
#    (filtered and unfiltered), a single-chunk index and an implicit index (early allocation, no filter)
x, y = arrays(12, 8, 'f4')
z = rng.standard_normal((6, 4)).astype('f8')
with h5py.File(os.path.join(OUT, 'latest_fixed_array.hdf5'), 'w', libver='latest') as f:
    f.create_dataset('input', data=x, chunks=(5, 1, 8, 8), compression='gzip', shuffle=True)
    f.create_dataset('output', data=y, chunks=(4, 3, 8, 4))
    f.create_dataset('single', data=z, chunks=(6, 4), compression='gzip')
    sp = h5py.h5s.create_simple((6, 4))
    pl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
    pl.set_chunk((4, 4))
    pl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
    did = h5py.h5d.create(f.id, b'implicit', h5py.h5t.NATIVE_DOUBLE, sp, pl)
    did.write(h5py.h5s.ALL, h5py.h5s.ALL, z)
    g = f.create_group('sub')
    g.create_dataset('input', data=x[:3])
done('latest_fixed_array.hdf5', {'input': x, 'output': y, 'single': z, 'implicit': z, 'sub/input': x[:3]})

# 5. latest, more than 1024 chunks: the fixed array's data block is PAGED
x = rng.standard_normal((1100, 1, 2, 2)).astype('f4')
with h5py.File(os.path.join(OUT, 'latest_fixed_array_paged.hdf5'), 'w', libver='latest') as f:
    f.create_dataset('input', data=x, chunks=(1, 1, 2, 2), compression='gzip')
    f.create_dataset('plain', data=x, chunks=(1, 1, 2, 2))
done('latest_fixed_array_paged.hdf5', {'input': x, 'plain': x})

# 6. big-endian floats and a user block (MATLAB-style 512-byte header)
x, y = arrays(4, 8)
with h5py.File(os.path.join(OUT, 'bigendian_userblock.hdf5'), 'w', userblock_size=512) as f:
    f.create_dataset('input', data=x.astype('>f4'))
    f.create_dataset('output', data=y.astype('>f8'), chunks=(2, 3, 8, 8), compression='gzip', shuffle=True)
done('bigendian_userblock.hdf5', {'input': x, 'output': y})

# 7. an extensible dataset (maxshape=None on the sample axis): v1 B-tree when libver is earliest
x, y = arrays(7, 8, 'f4')
with h5py.File(os.path.join(OUT, 'resizable_earliest.hdf5'), 'w') as f:
    d = f.create_dataset('input', shape=(3, 1, 8, 8), maxshape=(None, 1, 8, 8), dtype='f4', chunks=(2, 1, 8, 8), compression='gzip')
    d[:] = x[:3]
    d.resize(7, axis=0)
    d[3:] = x[3:]
done('resizable_earliest.hdf5', {'input': x})

# 8. the same under libver='latest': EXTENSIBLE ARRAY chunk index -- the reader must refuse it by name, not misread it
with h5py.File(os.path.join(OUT, 'resizable_latest.hdf5'), 'w', libver='latest') as f:
    d = f.create_dataset('input', shape=(7, 1, 8, 8), maxshape=(None, 1, 8, 8), dtype='f4', chunks=(2, 1, 8, 8), compression='gzip')
    d[:] = x
    f.create_dataset('output', data=y)
done('resizable_latest.hdf5', {'input': x, 'output': y})

with open(os.path.join(OUT, 'MANIFEST.json'), 'w') as f:
    json.dump(manifest, f, indent=1, sort_keys=True)
for k, v in sorted(manifest.items()):
    print(k, v['bytes'])
