"""CPU: the reference-order emulation the parity-mode HIP kernels run (regtr_amd/csrc/ref_umap.h + ref_kdtree.h, compiled for the host
by oracle/Makefile) against the real things: libstdc++'s std::unordered_map iteration order and std::sort, and the
unmodified reference C++ (nanoflann KD-tree visiting order + std::sort) through oracle/_ref."""
import ctypes
import os

import numpy as np
import pytest

from oracle import native
from tests.util import gold, synth_cloud

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), 'oracle', 'libref_order_host.so')
_u64p = ctypes.POINTER(ctypes.c_uint64)
_i32p = ctypes.POINTER(ctypes.c_int)
_f32p = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope='module')
def host():
    native.build(ref=False)
    lib = ctypes.CDLL(_SO)
    lib.emul_batch_neighbors.restype = ctypes.c_int
    lib.real_umap_growth.restype = ctypes.c_int
    lib.emul_umap_growth.restype = ctypes.c_int
    return lib


def _order(fn, keys):
    keys = np.ascontiguousarray(keys, np.uint64)
    out = np.empty(len(keys), np.int32)
    fn(keys.ctypes.data_as(_u64p), len(keys), out.ctypes.data_as(_i32p))
    return out


def test_umap_growth_schedule_matches_this_libstdcxx(host):
    a = np.zeros((64, 2), np.uint32); b = np.zeros((64, 2), np.uint32)
    na = host.real_umap_growth(3_000_000, a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 64)
    nb = host.emul_umap_growth(b.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 64)
    assert na >= 18 and nb >= na
    assert np.array_equal(a[:na], b[:na])


@pytest.mark.parametrize('m', [0, 1, 2, 13, 14, 15, 29, 30, 31, 100, 541, 542, 5087, 5088, 5089, 40000])
def test_umap_iteration_order(host, m):
    rng = np.random.default_rng(m)
    # voxel-like keys: small dense integers, and the size_t wrap-around values negative voxel offsets produce
    dense = rng.permutation(4 * m + 7)[:m].astype(np.uint64)
    for keys in (dense, dense * np.uint64(977) + np.uint64(2 ** 64 - 5000), rng.integers(0, 2 ** 63, m).astype(np.uint64) * np.uint64(2)):
        keys = np.unique(keys)[rng.permutation(len(np.unique(keys)))]
        assert np.array_equal(_order(host.emul_umap_order, keys), _order(host.real_umap_order, keys))


def _pack(d2, idx):
    return (d2.astype(np.float32).view(np.uint32).astype(np.uint64) << np.uint64(32)) | idx.astype(np.uint64)


@pytest.mark.parametrize('n', [0, 1, 2, 15, 16, 17, 18, 31, 33, 40, 64, 71, 100, 257, 1000])
def test_std_sort_with_ties(host, n):
    rng = np.random.default_rng(n)
    for levels in (3, 7, 40, 10 ** 6):                     # few distinct distances = many ties (the lattice case)
        for rep in range(20):
            d2 = rng.integers(0, levels, n).astype(np.float32) * np.float32(3.6e-5)
            v = _pack(d2, rng.permutation(n))
            a, b = v.copy(), v.copy()
            host.emul_sort(a.ctypes.data_as(_u64p), n)
            host.real_sort(b.ctypes.data_as(_u64p), n)
            assert np.array_equal(a, b)
    # organ-pipe / sorted / reversed inputs (median-of-three worst cases) and the heap-sort fallback on its own
    for d2 in (np.minimum(np.arange(n), n - 1 - np.arange(n)), np.arange(n), np.arange(n)[::-1], np.zeros(n)):
        v = _pack(np.asarray(d2, np.float32), np.arange(n))
        a, b = v.copy(), v.copy()
        host.emul_sort(a.ctypes.data_as(_u64p), n); host.real_sort(b.ctypes.data_as(_u64p), n)
        assert np.array_equal(a, b)
        a, b = v.copy(), v.copy()
        host.emul_heap_sort(a.ctypes.data_as(_u64p), n); host.real_heap_sort(b.ctypes.data_as(_u64p), n)
        assert np.array_equal(a, b)


def _emul_query(host, q, s, ql, sl, radius):
    q, s = np.ascontiguousarray(q, np.float32), np.ascontiguousarray(s, np.float32)
    ql, sl = np.ascontiguousarray(ql, np.int32), np.ascontiguousarray(sl, np.int32)
    out = _i32p()
    w = host.emul_batch_neighbors(q.ctypes.data_as(_f32p), len(q), s.ctypes.data_as(_f32p), len(s), ql.ctypes.data_as(_i32p),
                                  sl.ctypes.data_as(_i32p), len(ql), ctypes.c_float(radius), ctypes.byref(out))
    arr = np.ctypeslib.as_array(out, shape=(len(q) * w + 1,))[:len(q) * w].reshape(len(q), w).copy()
    host.emul_free(out)
    return arr


@pytest.mark.parametrize('case', ['modelnet', '3dmatch_crop'])
def test_neighbour_order_vs_reference_fixture(host, case):
    """Row-for-row, column-for-column equal to the tables the unmodified reference C++ produced (committed fixture):
    KD-tree visiting order + std::sort tie order included."""
    g = gold(f'native_{case}')
    pts, lens, r = g['pts'], g['lens'], float(g['radius'])
    assert np.array_equal(_emul_query(host, pts, pts, lens, lens, r), g['neighbors'])
    assert np.array_equal(_emul_query(host, g['sub_pts'], pts, g['sub_lens'], lens, r), g['pools'])


@pytest.mark.skipif(not native.have_ref(), reason='oracle/_ref not built (needs /root/reference)')
def test_neighbour_order_vs_unmodified_reference_cpp(host):
    rng = np.random.default_rng(11)
    clouds = [synth_cloud(rng, n, lattice=lat) for n, lat in ((3000, 0.006), (1, 0.0), (9, 0.0), (11, 0.006), (2500, 0.0), (800, 0.02))]
    clouds.append(np.repeat(synth_cloud(rng, 40), 12, axis=0))            # 12-fold duplicates: all-equal splits
    pts = np.concatenate(clouds); lens = np.array([len(c) for c in clouds], np.int32)
    for radius in (0.0625, 0.125, 0.3):
        ref = native.ref_batch_query(pts, pts, lens, lens, radius)
        assert np.array_equal(_emul_query(host, pts, pts, lens, lens, radius), ref)
    sub, sl = native.ref_subsample_batch(pts, lens, 0.05)
    assert np.array_equal(_emul_query(host, sub, pts, sl, lens, 0.0625), native.ref_batch_query(sub, pts, sl, lens, 0.0625))


@pytest.mark.skipif(not native.have_ref(), reason='oracle/_ref not built (needs /root/reference)')
@pytest.mark.parametrize('case', ['3dmatch_kitchen', '3dmatch_home_at'])
def test_neighbour_order_kitchen_pyramid(host, case):
    """Every table of the real red-kitchen / home_at pairs' pyramids (reference row order at every level; home_at is the dense one:
    22.7 % of level-0 rows are cut at K = 40, so which equidistant supports survive is exercised 2.5x as often)."""
    g = gold(case)
    pts = np.concatenate([g['src'], g['tgt']]); lens = np.array([len(g['src']), len(g['tgt'])], np.int32)
    r, dl = 0.0625, 0.05
    for l in range(3):
        assert np.array_equal(_emul_query(host, pts, pts, lens, lens, r), native.ref_batch_query(pts, pts, lens, lens, r))
        sub, sl = native.ref_subsample_batch(pts, lens, dl)
        assert np.array_equal(_emul_query(host, sub, pts, sl, lens, r), native.ref_batch_query(sub, pts, sl, lens, r))
        pts, lens, r, dl = sub, sl, r * 2, dl * 2
