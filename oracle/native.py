"""TEST INFRASTRUCTURE ONLY -- ctypes loaders for the two CPU checker libraries.

``oracle_*``  : our restatement, oracle/regtr_oracle.cpp  (always buildable: `make -C oracle oracle`)
``ref_*``     : the unmodified reference C++ behind oracle/ref_shim.cpp
                (oracle/_ref/libref_oracle.so; built where /root/reference exists, shipped prebuilt
                to the GPU box).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, 'libregtr_oracle.so')
_REF_SO = os.path.join(_HERE, '_ref', 'libref_oracle.so')

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(ref=True):
    subprocess.check_call(['make', '-s', '-C', _HERE, 'oracle'])
    if ref:
        subprocess.check_call(['make', '-s', '-C', _HERE, 'ref'])


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_oracle = None
_ref = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        if not os.path.exists(_ORACLE_SO) or \
                os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, 'regtr_oracle.cpp')):
            build(ref=False)
        _oracle = ctypes.CDLL(_ORACLE_SO)
        _oracle.oracle_grid_subsample.restype = ctypes.c_int
        _oracle.oracle_grid_subsample_keyed.restype = ctypes.c_int
    return _oracle


def have_ref():
    return os.path.exists(_REF_SO)


def ref_lib():
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError('oracle/_ref/libref_oracle.so not built (needs /root/reference)')
        _ref = ctypes.CDLL(_REF_SO)
        for f in ('ref_batch_neighbors', 'ref_batch_ordered_neighbors', 'ref_batch_grid_subsampling'):
            getattr(_ref, f).restype = ctypes.c_int
    return _ref


# ----------------------------------------------------------------------------- restatement
VOXEL_KEY_MODES = {'origin': 0, 'floor': 1, 'floor_rcp': 2}


def grid_subsample(points, lens, dl, return_keys=False, ref_order=False, key_mode=0):
    """-> (pts (M,3) f32, lens (B,) i32[, keys (M,) u64]).  Rows in canonical first-appearance order, or in
    the reference's libstdc++ unordered_map iteration order when ref_order=True.  key_mode 0: the CPU op's voxel rule;
    1 / 2: PreprocessorGPU's floor(p / dl) (regtr_oracle.cpp: oracle_grid_subsample_keyed)."""
    points, lens = _f32(points), _i32(lens)
    n, nb = points.shape[0], lens.shape[0]
    out = np.empty((max(n, 1), 3), np.float32)
    out_lens = np.empty(nb, np.int32)
    keys = np.empty(max(n, 1), np.uint64)
    m = oracle_lib().oracle_grid_subsample_keyed(
        points.ctypes.data_as(_f32p), n, lens.ctypes.data_as(_i32p), nb, ctypes.c_float(dl), int(ref_order), int(key_mode),
        out.ctypes.data_as(_f32p), out_lens.ctypes.data_as(_i32p),
        keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    if return_keys:
        return out[:m].copy(), out_lens, keys[:m].copy()
    return out[:m].copy(), out_lens


def radius_neighbors(queries, supports, q_lens, s_lens, radius, K):
    """-> (idx (Nq,K) i32 canonical (d2, index) order padded with Ns, count (Nq,) i32, tie (Nq,) bool)."""
    queries, supports, q_lens, s_lens = _f32(queries), _f32(supports), _i32(q_lens), _i32(s_lens)
    nq, ns = queries.shape[0], supports.shape[0]
    idx = np.empty((nq, K), np.int32)
    cnt = np.empty(nq, np.int32)
    tie = np.empty(nq, np.uint8)
    oracle_lib().oracle_radius_neighbors(
        queries.ctypes.data_as(_f32p), nq, supports.ctypes.data_as(_f32p), ns,
        q_lens.ctypes.data_as(_i32p), s_lens.ctypes.data_as(_i32p), q_lens.shape[0],
        ctypes.c_float(radius), K, idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p),
        tie.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return idx, cnt, tie.astype(bool)


# ----------------------------------------------------------------------------- unmodified reference
def ref_batch_query(queries, supports, q_lens, s_lens, radius, ordered=False):
    """cpp_neighbors.batch_query semantics: untruncated (Nq, max_count) int32, pad = Ns."""
    queries, supports, q_lens, s_lens = _f32(queries), _f32(supports), _i32(q_lens), _i32(s_lens)
    nq, ns = queries.shape[0], supports.shape[0]
    out = _i32p()
    fn = ref_lib().ref_batch_ordered_neighbors if ordered else ref_lib().ref_batch_neighbors
    w = fn(queries.ctypes.data_as(_f32p), nq, supports.ctypes.data_as(_f32p), ns,
           q_lens.ctypes.data_as(_i32p), s_lens.ctypes.data_as(_i32p), q_lens.shape[0],
           ctypes.c_float(radius), ctypes.byref(out))
    arr = np.ctypeslib.as_array(out, shape=(nq * w + 1,))[:nq * w].reshape(nq, w).copy()
    ref_lib().ref_free(out)
    return arr


def ref_subsample_batch(points, lens, sampleDl, max_p=0):
    """cpp_subsampling.subsample_batch semantics: (pts (M,3) f32 in the reference's order, lens (B,))."""
    points, lens = _f32(points), _i32(lens)
    out = _f32p()
    out_lens = np.empty(lens.shape[0], np.int32)
    m = ref_lib().ref_batch_grid_subsampling(
        points.ctypes.data_as(_f32p), points.shape[0], lens.ctypes.data_as(_i32p), lens.shape[0],
        ctypes.c_float(sampleDl), max_p, ctypes.byref(out), out_lens.ctypes.data_as(_i32p))
    arr = np.ctypeslib.as_array(out, shape=(3 * m + 1,))[:3 * m].reshape(m, 3).copy()
    ref_lib().ref_free(out)
    return arr, out_lens
