"""ctypes binding of libpdes_hip.so (include/pdes_hip.h).  There is NO fallback: if the HIP
library is missing or a call fails, the product path raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpdes_hip.so')
ABI_VERSION = 12

_c_f = ctypes.c_float
_c_i = ctypes.c_int
_c_p = ctypes.c_void_p

# name -> argtypes; mirrors include/pdes_hip.h one to one (tests check every symbol is exported)
SIGNATURES = {
    'pdes_abi_version': [],
    'pdes_stat_replicas': [],
    'pdes_darcy_loss': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_f, _c_i, _c_f, _c_f, _c_p],
    'pdes_sobel_grad': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p],
    'pdes_sobel_grad_adjoint': [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p],
    'pdes_conv_forward': [_c_p, _c_i, _c_p],
    'pdes_conv_backward_weight': [_c_p, _c_i, _c_p],
    'pdes_conv_backward_data': [_c_p, _c_i, _c_p],
    'pdes_backward': [_c_p, _c_i, _c_p, _c_p, _c_p, _c_p],
    'pdes_bn_backward_finalize': [_c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_i,
                                  ctypes.c_longlong, _c_p],
    'pdes_conv_wgrad_plan': [_c_p, _c_p, _c_p],
    'pdes_wgrad_reduce_all': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights_mfma': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights_up': [_c_p, _c_i, _c_i, _c_p],
    'pdes_pack_all': [_c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_p, _c_i, _c_i, _c_p],
    'pdes_pack_weights_b3': [_c_p, _c_i, _c_i, _c_p],
    'pdes_b3_image_elems': [_c_i, _c_i, _c_p, _c_p],
    'pdes_bn_update_running': [_c_p, _c_i, _c_i, _c_f, _c_i, ctypes.c_longlong, _c_p],
    'pdes_bn_param_grads': [_c_p, _c_i, _c_i, _c_i, ctypes.c_longlong, _c_p],
    'pdes_adam_step': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_f, ctypes.c_longlong, _c_p],
    'pdes_adam_step_host': [_c_p, _c_p, _c_p, _c_p, _c_p, _c_f, _c_i, ctypes.c_longlong, _c_p],
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


def check(rc, what):
    if rc != 0:
        msg = _ERR.get(rc, f'hipError_t {rc}' if rc > 0 else f'error {rc}')
        raise RuntimeError(f'{what} failed: {msg}')


def ptr(t):
    """device pointer of a torch tensor (None -> NULL); the tensor must be fp32/contiguous/cuda"""
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


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
