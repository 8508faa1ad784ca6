"""ctypes binding of libpdes_hip.so (include/pdes_hip.h).  There is NO fallback: if the HIP
library is missing or a call fails, the product path raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpdes_hip.so')
ABI_VERSION = 24

_c_f = ctypes.c_float
_c_i = ctypes.c_int
_c_p = ctypes.c_void_p

# name -> argtypes; mirrors include/pdes_hip.h one to one (tests check every symbol is exported)
SIGNATURES = {
    'pdes_abi_version': [],
    'pdes_sizeof': [_c_i],
    'pdes_stat_replicas': [],
    'pdes_context_create': [_c_p, _c_i],
    'pdes_context_destroy': [_c_p],
    'pdes_context_set_option': [_c_p, ctypes.c_char_p, ctypes.c_char_p],
    'pdes_context_load_env': [_c_p],
    'pdes_context_device': [_c_p],
    'pdes_darcy_loss': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_f, _c_p],
    'pdes_darcy_loss_dw': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p, _c_i, _c_f, _c_f, _c_p],
    'pdes_darcy_loss_partial_rows': [_c_i, _c_i, _c_i, _c_i],
    'pdes_sobel_grad': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p],
    'pdes_sobel_grad_adjoint': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p],
    'pdes_sobel5_grad': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p],
    'pdes_sobel5_grad_adjoint': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p],
    'pdes_conv_forward': [_c_p, _c_p, _c_i, _c_p],
    'pdes_conv_backward_weight': [_c_p, _c_p, _c_i, _c_p],
    'pdes_conv_backward_data': [_c_p, _c_p, _c_i, _c_p],
    'pdes_conv_image_use': [_c_p, _c_p, _c_p],
    'pdes_backward': [_c_p, _c_p, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p],
    'pdes_backward2': [_c_p, _c_p, _c_i, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p],
    'pdes_backward_chain': [_c_p, _c_p, _c_i, _c_i, _c_p],
    'pdes_backward_weights': [_c_p, _c_p, _c_i, _c_i, _c_p],
    'pdes_graph_begin': [_c_p],
    'pdes_graph_end': [_c_p, _c_p],
    'pdes_graph_nodes': [_c_p],
    'pdes_graph_launch': [_c_p, _c_p],
    'pdes_graph_destroy': [_c_p],
    'pdes_program_run': [_c_p, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p],
    'pdes_bn_backward_finalize': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_i,
                                  ctypes.c_longlong, _c_p],
    'pdes_conv_wgrad_plan': [_c_p, _c_p, _c_p, _c_p],
    'pdes_wgrad_reduce_all': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights_mfma': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights_up': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_all': [_c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_i, _c_p],
    'pdes_pack_all2': [_c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights_b3': [_c_p, _c_i, _c_i, _c_p],
    'pdes_b3_image_elems': [_c_i, _c_i, _c_p, _c_p],
    'pdes_pack_weights_b3up': [_c_p, _c_i, _c_i, _c_p],
    'pdes_b3up_image_elems': [_c_i, _c_i, _c_p, _c_p],
    'pdes_bn_update_running': [_c_p, _c_i, _c_i, _c_f, _c_i, ctypes.c_longlong, _c_p],
    'pdes_bn_param_grads': [_c_p, _c_i, _c_i, _c_i, ctypes.c_longlong, _c_p],
    'pdes_adam_step': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_f, ctypes.c_longlong, _c_p],
    'pdes_adam_step_host': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_i, ctypes.c_longlong, _c_p],
    'pdes_adam_step_host2': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_i, ctypes.c_longlong, _c_p, ctypes.c_longlong, _c_p],
    'pdes_test_metrics': [_c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p],
    'pdes_mse_partials': [ctypes.c_longlong],
    'pdes_mse_loss': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, ctypes.c_longlong, _c_p],
    'pdes_multi_dot': [_c_p, ctypes.c_longlong, _c_i, _c_p, _c_i, ctypes.c_longlong, _c_p, _c_i, _c_p],
    'pdes_flow_prepare': [_c_p, _c_i, _c_i, _c_p],
    'pdes_flow_param_grads': [_c_p, _c_i, _c_p, _c_i, _c_i, ctypes.c_longlong, _c_p],
    'pdes_flow_logp': [_c_p, _c_p, _c_i, _c_p, _c_i, _c_i, ctypes.c_longlong, _c_p],
    'pdes_step_tail': [_c_p, _c_i, _c_i, _c_f, _c_i, _c_p, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_p, _c_p, _c_i,
                       ctypes.c_longlong, _c_p],
}

_ERR = {-1: 'PDES_EINVAL (null pointer / non-positive size)',
        -2: 'PDES_ENOSUP (shape or option not implemented by the HIP kernels)',
        -3: 'PDES_EALIGN (pointer not 16-byte aligned)'}

_lib = None


def lib():
    """Load the library once.  Raises RuntimeError (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64: load it FIRST so this library binds to the same HIP
        # runtime (two runtimes in one process lose the device: hipErrorNoDevice)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -m pde_surrogate_amd.build` '
                '(there is no CPU / PyTorch fallback for the HIP path)')
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = _c_i
        v = L.pdes_abi_version()
        if v != ABI_VERSION:
            raise RuntimeError(f'libpdes_hip.so ABI {v} != binding ABI {ABI_VERSION}: rebuild')
        _lib = L
    return _lib


# ---- contexts: the library's only state (options + fork/join events), one per device, owned here ------------
N_EVENTS = 2048                     # pdes_backward with side streams needs n_descriptors + 4 (DenseED: 28 + 4; default cGlow: ~200)
import threading
_ctx_lock = threading.Lock()        # autograd's backward thread and the main thread may both ask for a context first
_contexts = {}                      # device index -> pdes_context*
_overrides = {}                     # option key -> value, applied to every context (set_option)

# signature of pdes_bucket_hook.fn (include/pdes_hip.h)
BUCKET_FN = ctypes.CFUNCTYPE(_c_i, _c_p, _c_i, _c_p)


class BucketHook(ctypes.Structure):
    _fields_ = [('fn', BUCKET_FN), ('user', _c_p)]


def context(device=None):
    """the pdes_context of a device (created on first use, on that device; the process environment's PDES_* knobs
    are read ONCE here, then `set_option` overrides apply)"""
    import torch
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    ctx = _contexts.get(idx)
    if ctx is None:
        with _ctx_lock:
            ctx = _contexts.get(idx)
            if ctx is None:
                L = lib()
                h = _c_p()
                with torch.cuda.device(idx):
                    check(L.pdes_context_create(ctypes.byref(h), N_EVENTS), 'pdes_context_create')
                check(L.pdes_context_load_env(h), 'pdes_context_load_env')
                for k, v in _overrides.items():
                    check(L.pdes_context_set_option(h, k.encode(), None if v is None else str(v).encode()), f'option {k}')
                ctx = _contexts[idx] = h
    return ctx


OPTIONS_EPOCH = 0


def set_option(key, value):
    """override a kernel-selection knob (include/pdes_hip.h lists the keys) on every context of this process;
    value None restores the compiled-in default"""
    global OPTIONS_EPOCH
    _overrides[key] = value
    OPTIONS_EPOCH += 1                  # (models re-query which weight images their kernels read: codec.py _pack_weights)
    for h in _contexts.values():
        check(lib().pdes_context_set_option(h, key.encode(), None if value is None else str(value).encode()),
              f'option {key}')


def destroy_contexts():
    for h in _contexts.values():
        lib().pdes_context_destroy(h)
    _contexts.clear()


def loss_partial_rows(B, H, W, flags=0):
    """rows of the `partials` workspace of pdes_darcy_loss for these arguments"""
    rows = lib().pdes_darcy_loss_partial_rows(B, H, W, flags)
    if rows <= 0:
        check(rows if rows < 0 else -2, 'pdes_darcy_loss_partial_rows')
    return rows


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f'hipError_t {rc}' if rc > 0 else f'error {rc}')
        raise RuntimeError(f'{what} failed: {msg}')


def ptr(t):
    """device pointer of a torch tensor (None -> NULL); the tensor must be fp32/contiguous/cuda"""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr(device=None):
    """the current HIP stream of `device` (default: the current device)"""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def device_guard(device):
    """make `device` current for the launches inside (streams, events and kernels of a call all belong to the device of
    its tensors); free when it already is"""
    import torch
    idx = torch.device(device).index
    if idx is None or idx == torch.cuda.current_device():
        return _NOGUARD
    return torch.cuda.device(idx)


def require_cuda(*tensors):
    import torch
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('pde_surrogate_amd runs on an MI355X only: got a CPU tensor '
                               '(there is no CPU fallback; move tensors to cuda)')
        if t.dtype != torch.float32 and t.dtype != torch.float64:
            raise RuntimeError(f'expected fp32 tensor, got {t.dtype}')
        if not t.is_contiguous():
            raise RuntimeError('expected a contiguous NCHW tensor')
