import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


_SEEDED = None


def load_seeded(module, prefix, buffers_prefix=None):
    """SURVEY 8(c) G6 fallback: install the seeded INITIAL values the reference-made fixtures started from
    (tests/golden/W_seeded.npz, written by tools/gen_golden.py round4 from the reference's own constructors under
    torch.manual_seed) into `module`, so that a torch whose RNG stream differs from the fixture generator's still runs
    every reference-pinned test instead of skipping it.  `prefix` names a full state_dict ('densed_seed1',
    'decoder_seed3') or, with `buffers_prefix`, parameters and buffers stored apart ('cglow_g19', 'cglow_g19_buffers').
    Every stored tensor must find its place (and every parameter of the module must be covered)."""
    import torch
    global _SEEDED
    if _SEEDED is None:
        _SEEDED = np.load(os.path.join(GOLDEN, 'W_seeded.npz'), allow_pickle=False)
    own = module.state_dict()
    seen = set()
    with torch.no_grad():
        for pre in filter(None, (prefix, buffers_prefix)):
            for k in _SEEDED.files:
                if k.startswith(pre + '/'):
                    name = k[len(pre) + 1:]
                    assert name in own, name
                    v = torch.from_numpy(_SEEDED[k])
                    assert tuple(own[name].shape) == tuple(v.shape), name
                    own[name].copy_(v.to(own[name].device))
                    seen.add(name)
    missing = [k for k, _ in module.named_parameters() if k not in seen]
    assert not missing, missing
    return module


def seeded_sd(prefix, like):
    """the stored seeded state_dict `prefix` as a dict of fresh tensors in the key order of `like` (oracle tests)"""
    import torch
    global _SEEDED
    if _SEEDED is None:
        _SEEDED = np.load(os.path.join(GOLDEN, 'W_seeded.npz'), allow_pickle=False)
    out = {k: torch.from_numpy(_SEEDED[prefix + '/' + k].copy()).reshape(like[k].shape) for k in like}
    assert len(out) == sum(f.startswith(prefix + '/') for f in _SEEDED.files)
    return out


def fixed_projection(shape, tag):
    """tools/gen_golden.py fixed_projection: the RNG-free direction a gradient tensor is projected on in G22"""
    n = int(np.prod(shape))
    return np.cos(0.37 * np.arange(n, dtype=np.float64) + 0.11 * tag).reshape(shape)


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _gpu_state():
    import torch
    from pde_surrogate_amd import _lib
    return torch.cuda.is_available(), os.path.exists(_lib.LIB_PATH)


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests skip ONLY on a machine without a GPU (this build container).  On a machine that HAS a GPU a missing
    libpdes_hip.so is an error, not a skip -- and with PDES_REQUIRE_GPU=1 (set it in the GPU-box command) so is a
    missing GPU: a green run can never be an all-skipped one."""
    gpu_items = [item for item in items if 'gpu' in item.keywords]
    if not gpu_items:
        return
    has_gpu, has_lib = _gpu_state()
    if has_gpu and has_lib:
        return
    if has_gpu and not has_lib:
        raise pytest.UsageError('a GPU is visible but pde_surrogate_amd/libpdes_hip.so is not built '
                                '(python -m pde_surrogate_amd.build): refusing to skip the GPU tests')
    if os.environ.get('PDES_REQUIRE_GPU', '0') not in ('', '0'):
        raise pytest.UsageError('PDES_REQUIRE_GPU=1 but torch sees no GPU: refusing to skip the GPU tests')
    skip = pytest.mark.skip(reason='no GPU on this machine (the -m gpu tests need an MI355X)')
    for item in gpu_items:
        item.add_marker(skip)


@pytest.fixture
def option():
    """set a kernel-selection option of the library (pdes_context_set_option) for one test; restored afterwards"""
    from pde_surrogate_amd import _lib
    touched = []

    def setter(key, value):
        touched.append(key)
        _lib.set_option(key, value)
    yield setter
    for k in touched:
        _lib.set_option(k, None)
        _lib._overrides.pop(k, None)


@pytest.fixture(autouse=True)
def _seed_every_test(request):
    """Every test starts from RNG streams that depend on ITS name only (round 6: inputs drawn from the global torch / numpy
    generators made a test's numbers depend on which tests ran before it -- with Adam's m / sqrt(v) behind them, enough to
    cross a bound in one suite order and not in another).  Tests that seed themselves are unaffected."""
    import zlib
    seed = zlib.crc32(request.node.nodeid.encode()) & 0x7fffffff
    np.random.seed(seed)
    try:
        import torch
        torch.manual_seed(seed)
    except ImportError:      # pragma: no cover
        pass
    yield
