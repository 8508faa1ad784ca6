"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/pdes_hip.h declares
(no compute calls without a GPU); the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def _declared():
    names = []
    for fn in os.listdir(os.path.join(ROOT, 'include')):
        if fn.endswith('.h'):
            txt = open(os.path.join(ROOT, 'include', fn)).read()
            txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
            names += re.findall(r'\bint\s+(pdes_\w+)\s*\(', txt)
    return names


def test_library_exports_every_declared_symbol():
    from pde_surrogate_amd import _lib, build
    build.build(verbose=False)
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert len(declared) >= 4
    for name in declared:
        assert hasattr(L, name), f'{name} declared in include/ but not exported'
    assert set(declared) == set(_lib.SIGNATURES), 'ctypes binding and header disagree'
    assert _lib.lib().pdes_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    from pde_surrogate_amd.models import darcy
    from pde_surrogate_amd.utils.image_gradient import SobelFilter
    K = torch.ones(1, 1, 64, 64)
    y = torch.zeros(1, 3, 64, 64, requires_grad=True)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        darcy.darcy_mixed_residual_loss(K, y)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        SobelFilter(64).grad_h(K)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under pde_surrogate_amd/ may reference it"""
    pkg = os.path.join(ROOT, 'pde_surrogate_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), os.path.join(d, f)


def test_ctypes_mirrors_have_the_size_of_the_c_structures():
    """the binding's ctypes mirrors of the boundary structures against sizeof() in the library (pdes_sizeof): a field
    added on one side only shifts every later field silently"""
    import ctypes
    from pde_surrogate_amd import _lib
    from pde_surrogate_amd.models import codec
    L = _lib.lib()
    mirrors = {0: codec.ConvDesc, 1: codec.PackItem, 2: codec.MfmaPackItem, 3: codec.UpPackItem, 4: codec.B3PackItem,
               5: codec.B3UpPackItem, 6: codec.ReduceItem, 7: codec.BnItem, 8: codec.PdesOp}
    for which, cls in mirrors.items():
        assert L.pdes_sizeof(which) == ctypes.sizeof(cls), (which, cls.__name__, L.pdes_sizeof(which), ctypes.sizeof(cls))
    assert L.pdes_sizeof(99) == -1
