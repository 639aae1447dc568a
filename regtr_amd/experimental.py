"""Measured-slower experiment kernels, kept reproducible but OUT of the product.

`libregtr_hip.so` does not contain them: csrc compiles them only under -DREGTR_EXPERIMENTAL, which `python -m regtr_amd.build
--experimental` (and __graft_entry__.build()) puts into a separate `libregtr_hip.experimental.so`.  Nothing in regtr_amd's forward imports
this module; tests/test_gpu_ops.py holds the kernels to the product path so the numbers in docs/NEGATIVES.md stay reproducible.
Entry points: include/regtr_hip_experimental.h (outside REGTR_ABI_VERSION).
"""
import ctypes
import os

import torch

from . import ops
from ._lib import _F, _I, _P, _Z, bptr, check, iptr, ptr, raw, stream

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libregtr_hip.experimental.so')

SIGNATURES = {
    'regtr_gemm_x3_strip_occupancy': (_I, [_I, _I, _I]),
    'regtr_kpconv_fused_supported': (_I, [_I, _I, _I, _I]),
    'regtr_kpconv_fused': (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _I, _F, _P, _P, _P]),
    'regtr_block_tail_res_supported': (_I, [_I, _I, _I]),
    'regtr_block_tail_res_ws_bytes': (_Z, [_I, _I, _I, _I]),
    'regtr_block_tail_res': (_I, [_P, _I, _P, _F, _P, _P, _I, _P, _P, _I, _I, _P, _I, _I, _I, _F, _F, _P, _I, _P, _Z, _P]),
}

_handle = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _handle
    if _handle is None:
        if not available():
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -m regtr_amd.build --experimental`')
        h = ctypes.PyDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype, fn.argtypes = res, args
        _handle = h
    return _handle


def kpconv_fused(q_xyz, nbr, x, xyzf, w_split, kernel_points, extent):
    """KPConv.forward (kpconv_blocks.py:269-414) in ONE launch for the level-0 shape (32 -> 32 channels): gather, kernel-point correlation,
    contraction and the division by the neighbour count, the weighted features staying in LDS.  3.35 ms against 2.09 + 0.99 ms for the two
    product kernels at level 0 of a 64-pair forward (docs/NEGATIVES.md, round 2)."""
    L = lib()
    nq, H = nbr.shape
    ns, Cin = x.shape
    KP = kernel_points.shape[0]
    assert L.regtr_kpconv_fused_supported(Cin, w_split.N, KP, H) and w_split.planes is not None
    out = torch.empty((nq, w_split.N), dtype=torch.float32, device=x.device)
    check(L.regtr_kpconv_fused(ptr(q_xyz), nq, ns, iptr(nbr), H, ptr(x), ptr(xyzf), ptr(kernel_points), KP, float(extent),
                               bptr(w_split.planes), ptr(out), stream()), 'regtr_kpconv_fused')
    return out


def block_tail_res(x1, x1_stats, sw1, res, res_stats, seg_off, max_len, slope=0.1, eps=1e-5):
    """LeakyReLU(InstanceNorm(x1' @ W1) + r), x1' = LeakyReLU(InstanceNorm(x1)) by x1_stats, r = res (identity / max-pooled shortcut,
    kpconv_blocks.py:734-741) or InstanceNorm(res) by res_stats (a Linear shortcut's product): unary2 is never written.  Level-1 blocks:
    moments 130 + prepare 68 + strip 410 us against strip GEMM 230 + normalise-add pass 340 us (docs/NEGATIVES.md, round 3)."""
    L = lib()
    M, K1 = x1.shape
    N = sw1.N
    n_clouds = seg_off.numel() - 1
    assert L.regtr_block_tail_res_supported(M, N, K1)
    nb = L.regtr_block_tail_res_ws_bytes(n_clouds, int(max_len), N, K1)
    ws = torch.empty(nb, dtype=torch.uint8, device=x1.device)
    ti = ops.tile_segments(seg_off, M, 256)
    y = torch.empty((M, N), dtype=torch.float32, device=x1.device)
    check(L.regtr_block_tail_res(raw(x1), x1.stride(0), ptr(x1_stats), slope, ptr(sw1.kn), raw(res), res.stride(0), ptr(res_stats),
                                 iptr(seg_off), n_clouds, int(max_len), iptr(ti), M, N, K1, eps, slope, ptr(y), N, bptr(ws), nb, stream()),
          'regtr_block_tail_res')
    return y
