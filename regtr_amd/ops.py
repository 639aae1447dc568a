"""Tensor-level wrappers over the C ABI (include/regtr_hip.h).  torch is used for device memory and the current
stream only; every arithmetic step runs in libregtr_hip.so.  Nothing here synchronises with the host, with one documented
exception: `KdTree` (the reference-order parity mode) reads its row width back.  Every pointer handed to the library is checked
for device, dtype and contiguity (_lib.ptr / iptr / bptr / dptr)."""
import math

import torch

from . import _lib, context, devflags
from ._lib import bptr, check, dptr, iptr, ptr, raw, stream


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------ preprocessing
F16_OPERAND_LIMIT = 65504.0 / 2     # f16 pair operands: largest finite f16 is 65504; weights are audited with a factor of headroom


VOXEL_KEY_MODES = {'origin': 0, 'floor': 1, 'floor_rcp': 2}


def grid_subsample(xyz, seg_off, n_cap, dl, row_order=0, key_mode=0, out_cap=None):
    """xyz (n_cap,3) f32, seg_off (B+1,) i32 [device] -> (out_xyz (n_cap,3) [first out_seg_off[-1] rows live],
    out_seg_off (B+1,) i32 [device]).  Row order: clouds stacked; voxels of a cloud by first appearance (row_order 0) or in
    the reference's libstdc++ unordered_map iteration order (row_order 1, parity mode).  key_mode: 0 the CPU op's voxel rule
    floor((p - origin) / dl), 1 PreprocessorGPU's floor(p / dl), 2 floor(p * (1 / dl)) (include/regtr_hip.h)."""
    L = _lib.lib()
    n_clouds = seg_off.numel() - 1
    out_cap = int(n_cap if out_cap is None else min(out_cap, n_cap))
    out = torch.empty((max(out_cap, 1), 3), dtype=torch.float32, device=xyz.device)
    out_off = torch.empty(n_clouds + 1, dtype=torch.int32, device=xyz.device)
    nb = L.regtr_grid_subsample_ordered_ws_bytes(n_cap, n_clouds, int(row_order))
    ws = _ws(nb, xyz.device)
    check(L.regtr_grid_subsample_ordered(ptr(xyz), iptr(seg_off), n_clouds, n_cap, float(dl), int(row_order), int(key_mode), out_cap, ptr(out),
                                         iptr(out_off), bptr(ws), nb, stream()), 'regtr_grid_subsample_ordered')
    return out, out_off


self_query_kernel = True     # tests / A-B runs: False routes self queries through the per-query kernel as well
SELF_QUERY_MIN_POINTS = 262144      # measured: 38k points (one pair) 48 us vs 15 us per table; 2.4M points (64 pairs) 1.25 vs 1.9 ms


class CellGrid:
    """Support-point cell grid for one radius; serves any number of radius queries (conv + pool tables of a level)."""

    def __init__(self, s_xyz, s_seg_off, ns_cap, radius):
        L = _lib.lib()
        self.n_clouds = s_seg_off.numel() - 1
        self.s_xyz, self.s_seg_off, self.ns_cap, self.radius = s_xyz, s_seg_off, int(ns_cap), float(radius)
        self.nbytes = L.regtr_cellgrid_ws_bytes(self.ns_cap, self.n_clouds)
        self.ws = _ws(self.nbytes, s_xyz.device)
        check(L.regtr_cellgrid_build(ptr(s_xyz), iptr(s_seg_off), self.n_clouds, self.ns_cap, self.radius, bptr(self.ws),
                                     self.nbytes, stream()), 'regtr_cellgrid_build')

    def query(self, q_xyz, q_seg_off, nq_cap, K, want_count=False, order=0):
        """order 0: the K nearest supports of the ball, rows ascending (d2, index) [reference CPU Preprocessor]; 1: the first K by support
        index, rows ascending by index [reference PreprocessorGPU / pytorch3d ball_query]."""
        L = _lib.lib()
        idx = torch.empty((max(nq_cap, 1), K), dtype=torch.int32, device=q_xyz.device)
        cnt = mx = None
        if want_count:
            cnt = torch.empty(max(nq_cap, 1), dtype=torch.int32, device=q_xyz.device)
            mx = torch.zeros(1, dtype=torch.int32, device=q_xyz.device)
        # the grid's own supports: cell-centric kernel (large sets only: one wave per query keeps more of the chip busy on a pair or two)
        if self_query_kernel and q_xyz is self.s_xyz and q_seg_off is self.s_seg_off and self.ns_cap >= SELF_QUERY_MIN_POINTS:
            check(L.regtr_radius_query_self(iptr(self.s_seg_off), self.ns_cap, self.n_clouds, self.radius, int(K), int(order),
                                            bptr(self.ws), self.nbytes, iptr(idx), iptr(cnt), iptr(mx), stream()), 'regtr_radius_query_self')
        else:
            check(L.regtr_radius_query(ptr(q_xyz), iptr(q_seg_off), int(nq_cap), iptr(self.s_seg_off), self.ns_cap,
                                       self.n_clouds, self.radius, int(K), int(order), bptr(self.ws), self.nbytes, iptr(idx), iptr(cnt),
                                       iptr(mx), stream()), 'regtr_radius_query')
        return (idx, cnt, mx) if want_count else idx


class KdTree:
    """Parity mode (cfg.kpconv_ref_row_order): nanoflann-order neighbour tables (csrc/ref_order.hip, in libregtr_parity.so: include/
    regtr_hip_parity.h).  Unlike CellGrid this synchronises with the host (the list capacity must cover the largest in-ball count, known only
    after a first pass)."""

    def __init__(self, s_xyz, s_seg_off, ns_cap):
        L = _lib.parity_lib()
        self.n_clouds = s_seg_off.numel() - 1
        self.s_xyz, self.s_seg_off, self.ns_cap = s_xyz, s_seg_off, int(ns_cap)
        self.nbytes = L.regtr_kdtree_ws_bytes(self.ns_cap, self.n_clouds)
        self.ws = _ws(self.nbytes, s_xyz.device)
        check(L.regtr_kdtree_build(ptr(s_xyz), iptr(s_seg_off), self.n_clouds, self.ns_cap, bptr(self.ws), self.nbytes, stream()),
              'regtr_kdtree_build')

    def query(self, q_xyz, q_seg_off, nq_cap, radius, K, list_cap=128):
        """-> (idx (nq_cap, K) i32 in the reference's row order, max in-ball count (host int))."""
        L = _lib.parity_lib()
        dev = q_xyz.device
        idx = torch.empty((max(nq_cap, 1), K), dtype=torch.int32, device=dev)
        while True:
            st = torch.zeros(2, dtype=torch.int32, device=dev)          # [max in-ball count, traversal-stack overflow]
            sb = L.regtr_kdtree_query_scratch_bytes(int(list_cap))
            scratch = _ws(sb, dev)
            check(L.regtr_kdtree_radius_query(ptr(q_xyz), iptr(q_seg_off), int(nq_cap), ptr(self.s_xyz), iptr(self.s_seg_off),
                                              self.ns_cap, self.n_clouds, float(radius), int(K), int(list_cap), bptr(self.ws),
                                              self.nbytes, bptr(scratch), sb, iptr(idx), None, st.data_ptr(), st.data_ptr() + 4,
                                              stream()), 'regtr_kdtree_radius_query')
            max_count, overflow = (int(v) for v in st.cpu())
            if overflow:
                raise RuntimeError('regtr_kdtree_radius_query: traversal stack overflow (tree deeper than the parity mode supports)')
            if max_count <= list_cap:
                break
            list_cap = max_count                 # a ball held more supports than the list: the sort needs them all, run again
        return idx, max_count


# ------------------------------------------------------------------------------------------------ dense
class SplitWeight:
    """A weight matrix prepared for both dense kernels: `kn` (K, N) float32 for regtr_gemm_f32 and, when N is a multiple
    of 64, the three bf16 planes of its exact 3-way split for regtr_gemm_x3.  Build once per parameter (cache it)."""

    def __init__(self, w, layout):
        """w: float32 CUDA tensor; layout 'nk' = (N, K) as nn.Linear stores it, 'kn' = (K, N)."""
        L = _lib.lib()
        w = w.detach().contiguous()
        if layout == 'nk':
            self.N, self.K = w.shape
            self.kn = w.t().contiguous()
        else:
            self.K, self.N = w.shape
            self.kn = w
        self.planes = None
        self._w, self._layout, self._planes16, self._f16_ok = w, layout, None, None
        if L.regtr_gemm_x3_supported(1, self.N, self.K) or L.regtr_gemm_stream_supported(1, self.N, self.K):
            self.planes = _ws(L.regtr_gemm_split_weights_bytes(self.N, self.K), w.device)
            check(L.regtr_gemm_split_weights(ptr(w), w.stride(0), self.N, self.K, 0 if layout == 'nk' else 1, bptr(self.planes),
                                             stream()), 'regtr_gemm_split_weights')


    @property
    def f16_ok(self):
        """The weights fit the f16 pair format's range (audited once per weight version: one small reduction + read-back; a weight
        at or beyond 65504 would make every product non-finite, so such a matrix stays on the bf16x3 planes)."""
        if self._f16_ok is None:
            self._f16_ok = bool(self._w.numel() == 0 or float(self._w.abs().max()) < F16_OPERAND_LIMIT)
        return self._f16_ok

    @property
    def planes16(self):
        """The f16 pair planes of regtr_gemm_split_weights_f16 (regtr_gemm_x3's n_planes = 4), built on first use."""
        if self._planes16 is None:
            L = _lib.lib()
            if not self.f16_ok:
                raise RuntimeError('SplitWeight.planes16: a weight of magnitude >= 65504 cannot take the f16 pair format (check .f16_ok)')
            self._planes16 = _ws(L.regtr_gemm_split_weights_f16_bytes(self.N, self.K), self._w.device)
            check(L.regtr_gemm_split_weights_f16(ptr(self._w), self._w.stride(0), self.N, self.K, 0 if self._layout == 'nk' else 1,
                                                 bptr(self._planes16), stream()), 'regtr_gemm_split_weights_f16')
        return self._planes16


# float32-grade contractions as three f16 MFMA terms (the f16 pair split, csrc/gemm_x3.hip) where the row-strip kernel serves the shape,
# instead of six (planes = 3) / three (planes = 2) bf16 terms.  f16_pair_default: what cfg.compute_dtype 'fp32' asks for (A-B runs:
# REGTR_DEV=1 REGTR_F16_PAIR=0).  Whether a given launch takes the format is a field of the per-forward context (context.current().f16_pair, set by
# RegTR.forward; `with ops.f16_pair(flag):` for tests and direct op calls) -- not a module global: two models on two host threads do not
# share it.
f16_pair_default = devflags.on('REGTR_F16_PAIR')
_f16_shape = {}


def f16_pair(on):
    """`with ops.f16_pair(True):` -- the enclosed launches (this thread) take the f16 pair format where it is served."""
    return context.current().derive(f16_pair=bool(on))


def f16_pair_ok(M, N, K, with_stats=False):
    key = (M, N, K, bool(with_stats))
    v = _f16_shape.get(key)
    if v is None:
        if len(_f16_shape) > 4096:
            _f16_shape.clear()
        v = _f16_shape[key] = bool(_lib.lib().regtr_gemm_x3_f16_supported(M, N, K, 1 if with_stats else 0))
    return v


class _GemmTimer:
    """bench.py's roofline_gemm: HIP events around one dense launch on its stream + what ran (context.recording(gemm_records=[...]))."""
    __slots__ = ('rec', 'e0', 'meta')

    def __init__(self, rec, **meta):
        self.rec, self.meta = rec, meta
        if rec is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def done(self, **more):
        if self.rec is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.rec.append((self.e0, e1, dict(self.meta, **more)))


_x3_shape = {}       # (M, N, K) -> (supported, preferred, workspace bytes, statistics tile rows, launch tile rows): host-side plan queries, memoised


def _x3_plan(M, N, K):
    p = _x3_shape.get((M, N, K))
    if p is None:
        L = _lib.lib()
        ok = bool(L.regtr_gemm_x3_supported(M, N, K))
        p = (ok, bool(ok and L.regtr_gemm_x3_preferred(M, N, K)), L.regtr_gemm_x3_ws_bytes(M, N, K) if ok else 0,
             (L.regtr_gemm_x3_stat_tile_rows(M, N, K) if M > 0 else 0) if ok else 0, L.regtr_gemm_x3_tile_rows(M, N, K) if ok else 0)
        if len(_x3_shape) > 4096:
            _x3_shape.clear()
        _x3_shape[(M, N, K)] = p
    return p


def gemm(a, b, bias=None, row_div=None, residual=None, relu=False, out=None, a_stats=None, a_seg_off=None, a_slope=0.1,
         want_stats=None, eps=1e-5, planes=3):
    """a (M,K) @ b with the fused epilogue of regtr_gemm_f32 / regtr_gemm_x3.  b: a (K,N) float32 tensor (exact-f32 MFMA
    kernel) or a SplitWeight (bf16x3 split kernel when the shape allows).  `a` may be a row-strided view.
    a_stats (n_seg,K,2) + a_seg_off: A is read as LeakyReLU(InstanceNorm(a)) (per-cloud stats) on the fly.
    want_stats = (seg_off, max_len): also return the per-cloud InstanceNorm (mean, rstd) table (n_clouds,N,2) of the
    result -- from the GEMM epilogue when the kernel supports it, else by a pass over the result.
    planes: bf16 planes per operand on the split kernel -- 3 float32-grade (default), 2 three-term split, 1 plain bf16."""
    L = _lib.lib()
    M, K = a.shape
    sw = b if isinstance(b, SplitWeight) else None
    b_kn = sw.kn if sw is not None else b
    Kb, N = b_kn.shape
    assert K == Kb and a.stride(1) == 1 and b_kn.is_contiguous()
    # the shallow encoder levels' Linears (short K, millions of rows, statistics wanted): one-shot strip kernel
    if (sw is not None and sw.planes is not None and want_stats is not None and bias is None and row_div is None and residual is None
            and not relu and planes == 3 and out is None and stream_ok(M, N, K, a, a_stats)
            and (a_stats is None or a_seg_off is want_stats[0])):
        tm = _GemmTimer(context.current().gemm_records, M=M, N=N, K=K, route='one-shot strip (bf16x3)', terms=6, w_bytes=6, fold=a_stats is not None, stats=True)
        res = gemm_stream(a, sw, want_stats[0], a_stats=a_stats, a_slope=a_slope, want_stats=True, eps=eps)
        tm.done()
        return res
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ldr = residual.stride(0) if residual is not None else 0
    lda = a.stride(0) if M > 1 else K
    ldc = out.stride(0) if M > 1 else N
    n_seg = a_seg_off.numel() - 1 if a_stats is not None else 0
    x3_ok, x3_pref, nb, x3_R, x3_rows = _x3_plan(M, N, K) if sw is not None and sw.planes is not None else (False, False, 0, 0, 0)
    # (N = 32, the level-0 KPConv contractions, stay on the exact-f32 kernel: a pure A stream on which the six-term bf16 strip loses, 1.20 vs
    #  1.14 ms, and the f16 pair strip ties -- 27.71 vs 27.73 ms per forward, docs/NEGATIVES.md)
    ctx = context.current()
    use_f16_pair = ctx.f16_pair and not ctx.force_x3
    if ctx.force_x3:
        planes = 3
    f16_range_log = ctx.f16_range_log
    if (x3_ok and lda % 4 == 0 and a.data_ptr() % 16 == 0 and not force_f32_gemm and (N >= 64 or a_stats is None)
            and (force_x3_gemm or x3_pref)):
        ws = _ws(nb, a.device) if nb else None
        R = x3_R if want_stats is not None else 0
        partial, s_off, n_clouds = None, None, 0
        if R:
            s_off = want_stats[0]
            n_clouds = s_off.numel() - 1
            partial = torch.empty((((M + R - 1) // R + n_clouds) * N, 2), dtype=torch.float64, device=a.device)
        seg_rows = s_off if R else a_seg_off                    # the cloud table of the rows, when the launch needs one
        ti = None
        if seg_rows is not None and use_tile_info and (a_seg_off is None or s_off is None or a_seg_off is s_off):
            ti = tile_segments(seg_rows, M, x3_rows)
        pl, npl = sw.planes, int(planes)
        if use_f16_pair and npl >= 2 and sw.f16_ok and f16_pair_ok(M, N, K, R > 0 or a_stats is not None or ti is not None):
            pl, npl = sw.planes16, 4
            if f16_range_log is not None:      # tests / audits: the largest operand magnitude handed to the f16 pair format (synchronises)
                f16_range_log.append((M, N, K, float(a.abs().max()) if M else 0.0, float(sw.kn.abs().max())))
        tm = _GemmTimer(ctx.gemm_records, M=M, N=N, K=K, route='split GEMM (f16 pair)' if npl == 4 else f'split GEMM (bf16 x{npl})',
                        terms={4: 3, 3: 6, 2: 3, 1: 1}[npl], w_bytes={4: 4, 3: 6, 2: 4, 1: 2}[npl], fold=a_stats is not None, stats=R > 0)
        check(L.regtr_gemm_x3(raw(a), lda, bptr(pl), raw(out), ldc, M, N, K, ptr(bias), ptr(row_div),
                              raw(residual), ldr, 1 if relu else 0,
                              ptr(a_stats), iptr(a_seg_off), n_seg, a_slope, bptr(ws), nb, dptr(partial), iptr(s_off), n_clouds,
                              npl, iptr(ti), ctx.status_ptr(), stream()), 'regtr_gemm_x3')
        tm.done()
        if R:
            stats = torch.empty((n_clouds, N, 2), dtype=torch.float32, device=a.device)
            check(L.regtr_instnorm_finalize_tiles(dptr(partial), iptr(s_off), n_clouds, N, R, eps, ptr(stats), stream()),
                  'regtr_instnorm_finalize_tiles')
            return out, stats
    else:
        nb = L.regtr_gemm_f32_ws_bytes(M, N, K)
        ws = _ws(nb, a.device) if nb else None
        tm = _GemmTimer(ctx.gemm_records, M=M, N=N, K=K, route='exact-f32 MFMA', terms=0, w_bytes=4, fold=a_stats is not None, stats=False)
        check(L.regtr_gemm_f32(raw(a), lda, ptr(b_kn), N, raw(out), ldc, M, N, K, ptr(bias), ptr(row_div),
                               raw(residual), ldr, 1 if relu else 0,
                               ptr(a_stats), iptr(a_seg_off), n_seg, a_slope, bptr(ws), nb, stream()), 'regtr_gemm_f32')
        tm.done()
    if want_stats is not None:
        return out, instnorm_stats(out, want_stats[0], want_stats[1], eps)
    return out


def tile_segments(seg_off, M, rows):
    """(n_tiles, 4) int32 table of regtr_tile_segments for the rows described by `seg_off`, cached on the offsets tensor itself (one
    tiny launch per pyramid level and tile height per forward)."""
    cache = getattr(seg_off, '_regtr_tiles', None)
    if cache is None:
        cache = seg_off._regtr_tiles = {}
    key = (M, rows, seg_off.data_ptr(), seg_off._version)      # an offsets tensor refilled in place must not serve stale cloud boundaries
    t = cache.get(key)
    if t is None:
        t = torch.empty(((M + rows - 1) // rows, 4), dtype=torch.int32, device=seg_off.device)
        check(_lib.lib().regtr_tile_segments(iptr(seg_off), seg_off.numel() - 1, M, rows, iptr(t), stream()), 'regtr_tile_segments')
        for k in [k for k in cache if k[2:] != key[2:]]:           # tables of the tensor's earlier contents
            del cache[k]
        cache[key] = t
    return t


def stream_ok(M, N, K, a, a_stats=None):
    """The one-shot strip kernel (csrc/gemm_stream.hip) serves this product and is the faster choice (tall problems only)."""
    return bool(use_stream_gemm and not force_f32_gemm and not force_x3_gemm and M >= STREAM_MIN_ROWS and a.stride(1) == 1
                and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0 and (a_stats is None or K <= 64)
                and _lib.lib().regtr_gemm_stream_supported(M, N, K))


def gemm_stream(a, sw, seg_off, a_stats=None, a_slope=0.1, want_stats=False, eps=1e-5):
    """regtr_gemm_stream: a' @ W for K in {32, 64, 128}, N <= 512 on millions of rows (see csrc/gemm_stream.hip).
    want_stats: also return the per-cloud InstanceNorm (mean, rstd) table of the result.  -> out | (out, stats)"""
    L = _lib.lib()
    M, K = a.shape
    N = sw.N
    n_clouds = seg_off.numel() - 1
    R = L.regtr_gemm_stream_tile_rows()
    ti = tile_segments(seg_off, M, R)
    partial = None
    if want_stats:
        partial = torch.empty((((M + R - 1) // R + n_clouds) * N, 2), dtype=torch.float64, device=a.device)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    check(L.regtr_gemm_stream(raw(a), a.stride(0) if M > 1 else K, bptr(sw.planes), ptr(out), N, M, N, K, ptr(a_stats), a_slope,
                              iptr(seg_off), n_clouds, iptr(ti), dptr(partial), stream()), 'regtr_gemm_stream')
    if not want_stats:
        return out
    stats = torch.empty((n_clouds, N, 2), dtype=torch.float32, device=a.device)
    check(L.regtr_instnorm_finalize_tiles(dptr(partial), iptr(seg_off), n_clouds, N, R, eps, ptr(stats), stream()),
          'regtr_instnorm_finalize_tiles')
    return out, stats


def block_tail_ok(x1, x1_stats, f, sw1, sw2):
    """regtr_block_tail serves this resnet-block tail (shape, alignment, size) and is switched on."""
    M, K1 = x1.shape
    return bool(use_block_tail and not force_f32_gemm and not force_x3_gemm and M >= STREAM_MIN_ROWS and f.shape[0] == M
                and x1_stats is not None and sw1.N == sw2.N and x1.stride(1) == 1 and f.stride(1) == 1
                and x1.stride(0) % 4 == 0 and f.stride(0) % 4 == 0 and x1.data_ptr() % 16 == 0 and f.data_ptr() % 16 == 0
                and x1_stats.data_ptr() % 16 == 0 and _lib.lib().regtr_block_tail_supported(M, sw1.N, K1, f.shape[1]))


def block_tail(x1, x1_stats, f, sw1, sw2, seg_off, max_len, slope=0.1, eps=1e-5, want_stats=False):
    """LeakyReLU(InstanceNorm(x1' @ W1) + InstanceNorm(f @ W2)), x1' = LeakyReLU(InstanceNorm(x1)) by x1_stats: the tail of a resnet
    bottleneck block with a Linear shortcut (kpconv_blocks.py:727-741) without writing either product (csrc/block_tail.hip).
    sw1 / sw2: SplitWeight of unary2 / unary_shortcut.  want_stats: also return the (2, n_clouds, N, 2) (mean, rstd) of the products."""
    return _block_tail(x1, x1_stats, None, f, sw1.kn, sw2.kn, seg_off, max_len, slope, eps, want_stats)


def _block_tail(x1, x1_stats, row_div, f, w1_kn, w2_kn, seg_off, max_len, slope, eps, want_stats):
    L = _lib.lib()
    M, K1 = x1.shape
    K2, N = (f.shape[1] if f is not None else 0), w1_kn.shape[1]
    n_clouds = seg_off.numel() - 1
    nb = L.regtr_block_tail_ws_bytes(n_clouds, int(max_len), N, K1, K2)
    ws = torch.empty(nb, dtype=torch.uint8, device=x1.device)       # not the shared scratch: sized per call, used across four launches
    ti = tile_segments(seg_off, M, 256)
    y = torch.empty((M, N), dtype=torch.float32, device=x1.device)
    st = torch.empty((2 if K2 else 1, n_clouds, N, 2), dtype=torch.float32, device=x1.device) if want_stats else None
    tm = _GemmTimer(context.current().gemm_records, M=M, N=N, K=K1 + K2, route='block tail (moments + two-source strip, bf16x3)', terms=6, w_bytes=6,
                    fold=True, stats=True, passes=2)
    check(L.regtr_block_tail(raw(x1), x1.stride(0), ptr(x1_stats), slope, ptr(row_div), raw(f) if f is not None else None,
                             f.stride(0) if f is not None else 0, ptr(w1_kn), ptr(w2_kn), iptr(seg_off), n_clouds, int(max_len),
                             iptr(ti), M, N, K1, K2, eps, slope, ptr(y), N, bptr(ws), nb, ptr(st), stream()), 'regtr_block_tail')
    tm.done()
    return (y, st) if want_stats else y


def first_block_ok(nq, Cin, KP, Cout):
    """kpconv_norm_lrelu serves the encoder's first block (one input feature, 15 kernel points, 64 outputs, a tall batch)."""
    return bool(use_block_tail and not force_f32_gemm and not force_x3_gemm and Cin == 1 and KP == 15 and nq >= STREAM_MIN_ROWS
                and _lib.lib().regtr_block_tail_supported(nq, Cout, 16, 0))


def kpconv_norm_lrelu(q_xyz, s_xyz, nbr, x, w16_kn, kernel_points, extent, seg_off, max_len, xyzf=None, slope=0.1, eps=1e-5,
                      want_stats=False):
    """SimpleBlock with one input feature (kpconv_blocks.py:590-646): KPConv -> InstanceNorm -> LeakyReLU.  The gather writes its
    15 weighted sums per query as 16-float rows; the contraction, the InstanceNorm (statistics from the rows' 16 x 16 second
    moments) and the LeakyReLU are one pass over them (regtr_block_tail, K2 = 0).  w16_kn: the (15, Cout) weights padded to (16, Cout)."""
    L = _lib.lib()
    nq, H = nbr.shape
    ns = x.shape[0]
    KP = kernel_points.shape[0]
    wf = torch.empty((nq, 16), dtype=torch.float32, device=x.device)
    num = torch.empty(nq, dtype=torch.float32, device=x.device)
    rec = context.current().gather_records           # bench.py's roofline: this gather is one of a forward's KPConv gather launches too
    if rec is not None:
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
    check(L.regtr_kpconv_gather(ptr(q_xyz), nq, ptr(s_xyz), ns, iptr(nbr), H, ptr(x), 1, None, ptr(xyzf), ptr(kernel_points), KP,
                                float(extent), None, None, 0, slope, ptr(wf), 16, ptr(num), stream()), 'regtr_kpconv_gather')
    if rec is not None:
        e1.record()
    res = _block_tail(wf, None, num, None, w16_kn, None, seg_off, max_len, slope, eps, want_stats)
    if rec is not None:
        e2.record()
        rec.append((e0, e1, e2, nq, H, 1, w16_kn.shape[1]))
    return res


STREAM_MIN_ROWS = 131072     # below this a forward is launch-bound (a pair or two): the tiled kernels and separate passes are as fast
# The SMALL-batch regime: forwards with fewer level-0 rows than this take the small-batch kernel forms at EVERY level (no packed support
# records / pre-normalised gather, no strip GEMM, no block tails) and the encoder's blocks go out through one C call (regtr_encoder_fwd).
# Measured with the one-call encoder (profiles/r05_ab_h.txt: bench.py --pairs n, small regime forced on / off): 2 pairs 3.01 vs 3.30 ms, 3 pairs
# 3.32 vs 3.50, 4 pairs 3.85 vs 3.67, 8 pairs 5.52 vs 5.33 -- the crossover sits between 114 k and 152 k rows.  (Round 4, op-by-op issue: 65536.)
SMALL_REGIME_ROWS = int(devflags.flag('REGTR_SMALL_ROWS', '131072'))
use_stream_gemm = devflags.on('REGTR_STREAM_GEMM')      # one-shot strip kernel for the shallow levels' Linears
use_block_tail = devflags.on('REGTR_BLOCK_TAIL')        # resnet-block tail / first block from input moments (csrc/block_tail.hip)
PRENORM_MIN_ROWS = 65536        # a LEVEL with fewer rows than this (of a large batch): the extra normalise pass costs more than the gather saves
prenorm_gather = devflags.on('REGTR_PRENORM')           # unary1's IN + LReLU applied before the gather (packed support records)
use_tile_info = devflags.on('REGTR_TILE_INFO')
# unary2's folded InstanceNorm + LeakyReLU operand applied by a separate in-place pass instead (kpconv.py ResnetBottleneckBlock): 1 = where the
# fold would route to the tiled kernel (K > 64), 2 = everywhere, 0 = never
preapply_unary2 = int(devflags.flag('REGTR_PREAPPLY_UNARY2', '1'))
PREAPPLY_MIN_ROWS = 8192        # (a pair or two per forward is launch-bound: the fold saves the extra launch there)
# csrc/encoder.hip sequences the small-batch regime with these rules compiled in (ENC_PREAPPLY_ROWS, no strip / block-tail form below
# STREAM_MIN_ROWS): the one-call path equals the op-by-op path only while the gates nest like this (kpconv.KPFEncoder._one_call_ok)
assert PREAPPLY_MIN_ROWS == 8192, 'csrc/encoder.hip compiles this gate in (ENC_PREAPPLY_ROWS)'      # (a REGTR_SMALL_ROWS override beyond STREAM_MIN_ROWS: _one_call_ok caps it)
# the six cross-encoder layers enqueued by one C call (regtr_cross_encoder_fwd) instead of 72 op calls
use_one_call_cross_encoder = devflags.on('REGTR_ONE_CALL_XENC')
# the encoder's blocks enqueued by one C call (regtr_encoder_fwd) in the small-batch regime (< 65536 level-0 rows) instead of ~130 op calls
use_one_call_encoder = devflags.on('REGTR_ONE_CALL_ENC')
force_f32_gemm = False      # tests / A-B runs: route every GEMM to the exact-f32 MFMA kernel
force_x3_gemm = False       # tests: route every supported shape to the split kernel, also where it is not the faster one


def layernorm(x, gamma, beta, add=None, eps=1e-5, want_plain=False, out=None):
    n, D = x.shape
    y = torch.empty_like(x) if out is None else out
    yp = torch.empty_like(x) if want_plain else None
    check(_lib.lib().regtr_layernorm(ptr(x), n, D, ptr(gamma), ptr(beta), eps, ptr(add), ptr(y), ptr(yp), stream()),
          'regtr_layernorm')
    return (y, yp) if want_plain else y


def add(a, b):
    """a + b, same shape, contiguous float32 (with_pos_embed of the post-norm encoder layer)."""
    assert a.shape == b.shape
    out = torch.empty_like(a)
    check(_lib.lib().regtr_add_f32(ptr(a), ptr(b), a.numel(), ptr(out), stream()), 'regtr_add_f32')
    return out


_posemb_tables = {}


def posemb_sine(xyz, d_model, scale=1.0, temperature=10000):
    """PositionEmbeddingCoordsSine.forward (position_embedding.py:29-50).  The 84-entry frequency table is
    init-time constant data computed exactly the way the reference computes it (float32 pow)."""
    n_dim = 3
    npf = d_model // n_dim // 2 * 2
    key = (npf, temperature, float(scale), xyz.device)
    ent = _posemb_tables.get(key)
    if ent is None:          # (once per configuration and device: building these per forward cost a one-pair forward 0.2 ms of host time)
        dim_t = torch.arange(npf, dtype=torch.float32)
        dim_t = (temperature ** (2 * torch.div(dim_t, 2, rounding_mode='trunc') / npf)).to(xyz.device)
        ent = _posemb_tables[key] = (dim_t, torch.tensor(scale * 2 * math.pi, dtype=torch.float32).item())
    dim_t, scale32 = ent
    pe = torch.empty((xyz.shape[0], d_model), dtype=torch.float32, device=xyz.device)
    check(_lib.lib().regtr_posemb_sine(ptr(xyz), xyz.shape[0], npf, d_model, scale32, ptr(dim_t), ptr(pe), stream()),
          'regtr_posemb_sine')
    return pe


# ------------------------------------------------------------------------------------------------ KPConv encoder
def kpconv(q_xyz, s_xyz, nbr, x, w_flat, kernel_points, extent, x_stats=None, s_seg_off=None, q_seg_off=None, slope=0.1,
           want_stats=None, xyzf=None):
    """KPConv.forward (kpconv_blocks.py:269-414), non-deformable / linear / sum.
    nbr (Nq,H) i32, x (Ns,Cin), w_flat (KP*Cin, Cout) -> (Nq, Cout).
    x_stats (n_clouds,Cin,2): the input features are LeakyReLU(InstanceNorm(x)) applied on the fly (the tail of the
    preceding UnaryBlock, kpconv_blocks.py:556-561), with s_seg_off / q_seg_off the support / query cloud offsets.
    xyzf (Ns,4): (x, y, z, f) records of the supports, f = row-positivity of x as instnorm_apply(row_xyz=, row_positive=) writes
    them (x final, no x_stats), or the feature itself when Cin == 1: one 16-byte load per neighbour in the gather.
    want_stats = (seg_off, max_len) of the QUERY rows: returns (out, InstanceNorm stats of out)."""
    L = _lib.lib()
    assert xyzf is None or x_stats is None
    nq, H = nbr.shape
    ns, Cin = x.shape
    KP = kernel_points.shape[0]
    dev = x.device
    n_seg = s_seg_off.numel() - 1 if x_stats is not None else 0
    # the matrix-core gather derives the positivity flags from the rows it reads; every other case (channel counts / row
    # widths it does not take, unaligned views, >= 2^29 feature elements, no supports) needs them precomputed
    fused_flag = (L.regtr_kpconv_gather_computes_flag(Cin, H) and x.data_ptr() % 16 == 0 and ns > 0 and ns * Cin < (1 << 29)
                  and (x_stats is None or x_stats.data_ptr() % 16 == 0))
    flag = None
    if not fused_flag:
        xyzf = None
        flag = torch.empty(ns, dtype=torch.float32, device=dev)
        check(L.regtr_rowsum_positive(ptr(x), ns, Cin, ptr(x_stats), iptr(s_seg_off) if x_stats is not None else None, n_seg,
                                      slope, ptr(flag), stream()), 'regtr_rowsum_positive')
    wf = torch.empty((nq, KP * Cin), dtype=torch.float32, device=dev)
    num = torch.empty(nq, dtype=torch.float32, device=dev)
    rec = context.current().gather_records
    if rec is not None:
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
    check(L.regtr_kpconv_gather(ptr(q_xyz), nq, ptr(s_xyz), ns, iptr(nbr), H, ptr(x), Cin, ptr(flag), ptr(xyzf),
                                ptr(kernel_points), KP, float(extent), ptr(x_stats),
                                iptr(q_seg_off) if x_stats is not None else None, n_seg, slope, ptr(wf), 0, ptr(num), stream()),
          'regtr_kpconv_gather')
    if rec is not None:
        e1.record()
    res = gemm(wf, w_flat, row_div=num, want_stats=want_stats)       # (out, stats) when want_stats is given
    if rec is not None:
        e2.record()
        rec.append((e0, e1, e2, nq, H, Cin, (res[0] if want_stats is not None else res).shape[1]))
    return res


# (bench.py times every KPConv gather launch with HIP events on the launch stream: context.recording(gather_records=[...]))


def maxpool(x, nbr, width=None):
    """max_pool (kpconv_blocks.py:127-143).  width: use only the first `width` columns of nbr -- the reference's CPU tables are
    min(max in-ball count, K) wide (kpconv.py:255-258), so a full row holds no zero shadow row (parity mode)."""
    ns, C = x.shape
    nq, H = nbr.shape
    out = torch.empty((nq, C), dtype=torch.float32, device=x.device)
    check(_lib.lib().regtr_maxpool_gather(ptr(x), ns, C, iptr(nbr), H, nq, H if width is None else int(width), ptr(out), stream()),
          'regtr_maxpool_gather')
    return out


def instnorm_stats(x, seg_off, max_len, eps=1e-5):
    L = _lib.lib()
    n_clouds = seg_off.numel() - 1
    C = x.shape[1]
    stats = torch.empty((n_clouds, C, 2), dtype=torch.float32, device=x.device)
    nb = L.regtr_instnorm_ws_bytes(n_clouds, max_len, C)
    ws = _ws(nb, x.device)
    check(L.regtr_instnorm_stats(ptr(x), iptr(seg_off), n_clouds, int(max_len), C, eps, ptr(stats), bptr(ws), nb,
                                 stream()), 'regtr_instnorm_stats')
    return stats


def instnorm_apply(x, seg_off, max_len, stats, residual=None, res_stats=None, lrelu=False, slope=0.1, out=None, row_positive=None,
                   row_xyz=None):
    """row_positive: optional (rows,) float32 output, 1.0 where the result's row sum is > 0 (KPConv's normaliser flag); with
    row_xyz (rows,3) it is (rows,4) and receives (x, y, z, flag) records -- the `xyzf` operand of kpconv()."""
    assert row_xyz is None or (row_positive is not None and row_positive.shape == (x.shape[0], 4) and row_xyz.shape == (x.shape[0], 3))
    n_clouds = seg_off.numel() - 1
    C = x.shape[1]
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().regtr_instnorm_apply(ptr(x), iptr(seg_off), n_clouds, int(max_len), C, ptr(stats), ptr(residual),
                                          ptr(res_stats), 1 if lrelu else 0, slope, ptr(out), ptr(row_xyz), ptr(row_positive), stream()),
          'regtr_instnorm_apply')
    return out


# ------------------------------------------------------------------------------------------------ attention + pose
def mha(q, k, v, seg_off, kv_of, max_len, n_heads, precision=0):
    """q, k, v: (N, E) column views of a packed projection; returns (N, E) concatenated heads.
    precision: 0 float32-grade (bf16x3 split MFMA), 1 plain bf16 operands, 2 exact-f32 MFMA (see regtr_hip.h)."""
    N, E = q.shape
    hd = E // n_heads
    out = torch.empty((N, E), dtype=torch.float32, device=q.device)
    ctx = context.current()
    rec = ctx.mha_records
    if ctx.force_x3 and precision == 3:
        precision = 0                      # the range fallback: bf16x3 operands (float32's range)
    if rec is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().regtr_mha_fwd(raw(q), q.stride(0), raw(k), k.stride(0), raw(v), v.stride(0),
                                   ptr(out), E, iptr(seg_off), iptr(kv_of), seg_off.numel() - 1, int(max_len), n_heads, hd,
                                   1.0 / math.sqrt(hd), int(precision), ctx.status_ptr(), stream()), 'regtr_mha_fwd')
    if rec is not None:
        e1.record()
        rec.append((e0, e1))
    return out



def attn_xyz(q, k, xyz, seg_off, kv_of, max_len):
    """CorrespondenceDecoder.simple_attention: q, k (L, N, D) contiguous, xyz (N, 3) -> (L, N, 3)."""
    Lyr, N, D = q.shape
    out = torch.empty((Lyr, N, 3), dtype=torch.float32, device=q.device)
    check(_lib.lib().regtr_attn_xyz(ptr(q), ptr(k), ptr(xyz), ptr(out), iptr(seg_off), iptr(kv_of), seg_off.numel() - 1, N, Lyr,
                                    int(max_len), D, 1.0 / math.sqrt(D), stream()), 'regtr_attn_xyz')
    return out


def weighted_procrustes(kp, corr, logit, seg_off, n_pairs):
    """kp (N,3), corr (L,N,3), logit (L,N), seg_off (2B+1,) i32 -> pose (L,B,3,4)."""
    Lyr, N = logit.shape
    pose = torch.empty((Lyr, n_pairs, 3, 4), dtype=torch.float32, device=kp.device)
    check(_lib.lib().regtr_weighted_procrustes(ptr(kp), ptr(corr), ptr(logit), iptr(seg_off), n_pairs, N, Lyr, ptr(pose),
                                               context.current().status_ptr(), stream()), 'regtr_weighted_procrustes')
    return pose
