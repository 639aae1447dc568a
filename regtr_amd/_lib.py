"""ctypes binding of libregtr_hip.so (the C ABI declared in include/regtr_hip.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails this raises, so a GPU run can never
silently pass on some other code path.
"""
import ctypes
import os
import threading

import torch

from . import context

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libregtr_hip.so')
if os.environ.get('REGTR_DEV', '') == '1' and os.environ.get('REGTR_VARIANT'):      # development only: A/B kernel experiments (regtr_amd/build.py)
    LIB_PATH = os.path.join(_HERE, f"libregtr_hip.{os.environ['REGTR_VARIANT']}.so")

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int
_F = _c.c_float
_Z = _c.c_size_t

class Weight(_c.Structure):
    """regtr_weight_t (include/regtr_hip.h)."""
    _fields_ = [('kn', _P), ('planes', _P), ('planes16', _P), ('N', _I), ('K', _I)]


class EncoderBlock(_c.Structure):
    """regtr_encoder_block_t."""
    _fields_ = [('kind', _I), ('strided', _I), ('layer', _I), ('n_kp', _I), ('extent', _F), ('kernel_points', _P),
                ('unary1', Weight), ('conv', Weight), ('unary2', Weight), ('shortcut', Weight)]


class EncoderLevel(_c.Structure):
    """regtr_encoder_level_t."""
    _fields_ = [('points', _P), ('n', _I), ('conv_idx', _P), ('pool_idx', _P), ('K', _I), ('pool_width', _I), ('seg_off', _P), ('max_len', _I)]


# name -> (restype, argtypes); mirrors include/regtr_hip.h one to one
ABI_VERSION = 11         # REGTR_ABI_VERSION of the include/regtr_hip.h these signatures mirror

# libregtr_parity.so (include/regtr_hip_parity.h): parity mode's KD-tree-order neighbour search, loaded on first use (parity_lib)
PARITY_SIGNATURES = {
    'regtr_parity_abi_version': (_I, []),
    'regtr_kdtree_ws_bytes': (_Z, [_I, _I]),
    'regtr_kdtree_query_scratch_bytes': (_Z, [_I]),
    'regtr_kdtree_build': (_I, [_P, _P, _I, _I, _P, _Z, _P]),
    'regtr_kdtree_radius_query': (_I, [_P, _P, _I, _P, _P, _I, _I, _F, _I, _I, _P, _Z, _P, _Z, _P, _P, _P, _P, _P]),
}

SIGNATURES = {
    'regtr_abi_version': (_I, []),
    'regtr_grid_subsample_ws_bytes': (_Z, [_I, _I]),
    'regtr_grid_subsample': (_I, [_P, _P, _I, _I, _F, _P, _P, _P, _Z, _P]),
    'regtr_grid_subsample_ordered_ws_bytes': (_Z, [_I, _I, _I]),
    'regtr_grid_subsample_ordered': (_I, [_P, _P, _I, _I, _F, _I, _I, _I, _P, _P, _P, _Z, _P]),
    'regtr_cellgrid_ws_bytes': (_Z, [_I, _I]),
    'regtr_cellgrid_build': (_I, [_P, _P, _I, _I, _F, _P, _Z, _P]),
    'regtr_radius_query': (_I, [_P, _P, _I, _P, _I, _I, _F, _I, _I, _P, _Z, _P, _P, _P, _P]),
    'regtr_radius_query_self': (_I, [_P, _I, _I, _F, _I, _I, _P, _Z, _P, _P, _P, _P]),
    'regtr_nearest_in_radius': (_I, [_P, _P, _I, _P, _I, _I, _c.c_double, _F, _P, _Z, _P, _P]),
    'regtr_overlap_avgpool': (_I, [_P, _I, _P, _I, _I, _I, _P, _P]),
    'regtr_rowsum_positive': (_I, [_P, _I, _I, _P, _P, _I, _F, _P, _P]),
    'regtr_kpconv_gather_computes_flag': (_I, [_I, _I]),
    'regtr_kpconv_gather': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _F, _P, _P, _I, _F, _P, _I, _P, _P]),
    'regtr_maxpool_gather': (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _P]),
    'regtr_instnorm_ws_bytes': (_Z, [_I, _I, _I]),
    'regtr_instnorm_stats': (_I, [_P, _P, _I, _I, _I, _F, _P, _P, _Z, _P]),
    'regtr_instnorm_apply': (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _I, _F, _P, _P, _P, _P]),
    'regtr_gemm_f32_ws_bytes': (_Z, [_I, _I, _I]),
    'regtr_gemm_f32': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _F, _P, _Z, _P]),
    'regtr_gemm_x3_supported': (_I, [_I, _I, _I]),
    'regtr_gemm_x3_preferred': (_I, [_I, _I, _I]),
    'regtr_gemm_split_weights_bytes': (_Z, [_I, _I]),
    'regtr_gemm_split_weights': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'regtr_gemm_x3_f16_supported': (_I, [_I, _I, _I, _I]),
    'regtr_gemm_split_weights_f16_bytes': (_Z, [_I, _I]),
    'regtr_gemm_split_weights_f16': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'regtr_gemm_x3_ws_bytes': (_Z, [_I, _I, _I]),
    'regtr_gemm_x3': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _I, _F, _P, _Z, _P, _P, _I, _I, _P, _P, _P]),
    'regtr_tile_segments': (_I, [_P, _I, _I, _I, _P, _P]),
    'regtr_gemm_x3_tile_rows': (_I, [_I, _I, _I]),
    'regtr_gemm_stream_supported': (_I, [_I, _I, _I]),
    'regtr_gemm_stream_tile_rows': (_I, []),
    'regtr_block_tail_supported': (_I, [_I, _I, _I, _I]),
    'regtr_block_tail_ws_bytes': (_Z, [_I, _I, _I, _I, _I]),
    'regtr_block_tail': (_I, [_P, _I, _P, _F, _P, _P, _I, _P, _P, _P, _I, _I, _P, _I, _I, _I, _I, _F, _F, _P, _I, _P, _Z, _P, _P]),
    'regtr_gemm_stream': (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _F, _P, _I, _P, _P, _P]),
    'regtr_gemm_x3_stat_tile_rows': (_I, [_I, _I, _I]),
    'regtr_instnorm_finalize_tiles': (_I, [_P, _P, _I, _I, _I, _F, _P, _P]),
    'regtr_add_f32': (_I, [_P, _P, _Z, _P, _P]),
    'regtr_layernorm': (_I, [_P, _I, _I, _P, _P, _F, _P, _P, _P, _P]),
    'regtr_posemb_sine': (_I, [_P, _I, _I, _I, _F, _P, _P, _P]),
    'regtr_mha_fwd': (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P]),
    'regtr_cross_encoder_per_layer_params': (_I, []),
    'regtr_cross_encoder_supported': (_I, [_I, _I, _I, _I]),
    'regtr_cross_encoder_ws_bytes': (_Z, [_I, _I, _I]),
    'regtr_cross_encoder_fwd': (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _Z, _P, _P, _P]),
    'regtr_encoder_supported': (_I, [_P, _I, _P, _I, _I, _I, _I]),
    'regtr_encoder_ws_bytes': (_Z, [_P, _I, _P, _I, _I, _I, _I, _I]),
    'regtr_encoder_fwd': (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _P, _I, _F, _F, _P, _Z, _P, _P]),
    'regtr_attn_xyz': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P]),
    'regtr_weighted_procrustes': (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
}

COMPOSITE = ('regtr_encoder_fwd', 'regtr_cross_encoder_fwd')      # bound through a GIL-releasing handle (see _load)

_ERR = {-1: 'kernel launch failed', -2: 'invalid argument', -3: 'workspace too small'}

_lib = None
_load_lock = threading.Lock()


def lib():
    global _lib
    if _lib is None:
        with _load_lock:
            if _lib is None:
                _lib = _load()
    return _lib


def _load():
    _lib = None
    if True:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -m regtr_amd.build` '
                               '(hipcc --offload-arch=gfx950). There is no CPU fallback.')
        # PyDLL: the calls keep the GIL.  Every entry point only ENQUEUES work (microseconds); releasing the GIL around ~280 such calls per
        # forward makes the launching thread re-acquire it 280 times, and with any other busy Python thread in the process (the loader's
        # upload thread, a second model) each re-acquisition can wait a full switch interval (5 ms): 50-100 ms stalls per forward were
        # measured that way (profiles/r04_e2e_harness.txt)
        _lib = ctypes.PyDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.restype = res
            fn.argtypes = args
        # ... except the COMPOSITE entry points, which enqueue ~70-130 launches per call (0.3-0.5 ms inside the library, and hipLaunchKernel
        # can block on queue back-pressure): those release the GIL (a CDLL handle of the same library), so a loader / upload thread or a
        # second model's thread is not frozen for the duration.  The trade-off for the short calls stays as described above.
        nogil = ctypes.CDLL(LIB_PATH)
        for name in COMPOSITE:
            fn = getattr(nogil, name)
            fn.restype, fn.argtypes = SIGNATURES[name]
            setattr(_lib, name, fn)
        got = _lib.regtr_abi_version()
        if got != ABI_VERSION:
            _lib = None
            raise RuntimeError(f'{LIB_PATH} implements C-ABI version {got}, this binding was written for {ABI_VERSION}: rebuild with '
                               '`python -m regtr_amd.build --force`')
    return _lib


_parity = None
PARITY_LIB_PATH = os.path.join(_HERE, 'libregtr_parity.so')


def parity_lib():
    """libregtr_parity.so, loaded when parity mode (cfg.kpconv_ref_row_order / cpp_wrappers.reference_order) first asks for a KD-tree-order
    neighbour table; the default path never does.  Missing library -> RuntimeError (there is no fallback)."""
    global _parity
    if _parity is None:
        with _load_lock:
            if _parity is None:
                if not os.path.exists(PARITY_LIB_PATH):
                    raise RuntimeError(f'{PARITY_LIB_PATH} is missing: build it with `python -m regtr_amd.build` (parity mode only)')
                lib_ = ctypes.PyDLL(PARITY_LIB_PATH)
                for name, (res, args) in PARITY_SIGNATURES.items():
                    fn = getattr(lib_, name)
                    fn.restype, fn.argtypes = res, args
                if lib_.regtr_parity_abi_version() != ABI_VERSION:
                    raise RuntimeError(f'{PARITY_LIB_PATH} was built against C-ABI version {lib_.regtr_parity_abi_version()}, this binding is {ABI_VERSION}: '
                                       'rebuild with `python -m regtr_amd.build --force`')
                _parity = lib_
    return _parity


def check(status, what):
    if status != 0:
        raise RuntimeError(f'{what}: {_ERR.get(status, "error")} (status {status})')


on_device = context.on_device     # `with on_device(dev):` pins the launch device for the enclosed op calls (thread-local, regtr_amd/context.py)


def ptr(t, dtype=torch.float32):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous, of the dtype the kernel reads (float32
    unless stated) and live on the GPU the launch goes to.  (Called a few thousand times per forward: three C-level calls.)"""
    if t is None:
        return None
    dev = context.current().device_index
    if t.get_device() != (dev if dev is not None else torch.cuda.current_device()) or t.dtype is not dtype or not t.is_contiguous():
        _explain(t, dtype)
    return t.data_ptr()


def _explain(t, dtype):
    if not t.is_cuda:
        raise RuntimeError('regtr_amd ops need GPU tensors (no CPU fallback)')
    if t.dtype != dtype:
        raise RuntimeError(f'regtr_amd op expected a {dtype} tensor, got {t.dtype} (the kernels reinterpret nothing)')
    if not t.is_contiguous():
        raise RuntimeError('regtr_amd ops need contiguous tensors')
    dev = context.current().device_index
    raise RuntimeError(f'tensor on {t.device} but the launch device is cuda:{dev if dev is not None else torch.cuda.current_device()}; '
                       'wrap the call in regtr_amd._lib.on_device(tensor.device)')


def iptr(t):
    return ptr(t, torch.int32)


def bptr(t):
    return ptr(t, torch.uint8)


def dptr(t):
    return ptr(t, torch.float64)


def raw(t):
    """data_ptr of a float32 row-strided view (unit column stride is the caller's business)."""
    if t is None:
        return None
    if t.dtype is not torch.float32 or t.get_device() < 0:
        raise RuntimeError(f'regtr_amd op expected a float32 GPU tensor, got {t.dtype} on {t.device}')
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """hipStream_t (as an int) of torch's current stream on the launch device.  ~200 launches per forward ask: the raw-handle C call
    costs 0.2 us where torch.cuda.current_stream() builds a Stream object through four Python frames (8 us; it was the largest single
    host cost of a one-pair forward, tools/host_profile.py)."""
    dev = context.current().device_index
    if _raw_stream is not None:
        return _raw_stream(dev if dev is not None else torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
