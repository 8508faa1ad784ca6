"""Build libpdes_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

    python -m pde_surrogate_amd.build            # rebuild if any source is newer than the .so
    python -m pde_surrogate_amd.build --force

The .so is git-ignored (history stays source-only) but travels with the gpurun snapshot.
hipcc cross-compiles gfx950 without a GPU.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpdes_hip.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=fast',
         '-Wno-unused-result', '-munsafe-fp-atomics']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc')) + glob.glob(os.path.join(os.path.dirname(HERE), 'include', '*.h'))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in sources() + headers())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    # one object per source, compiled in parallel, then linked: keeps rebuilds to seconds
    procs = []
    for src in sources():
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(
                [os.path.getmtime(src)] + [os.path.getmtime(h) for h in headers()]):
            cmd = [hipcc] + [f for f in FLAGS if f != '-shared'] + os.environ.get('PDES_EXTRA_FLAGS', '').split() + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {src}')
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
