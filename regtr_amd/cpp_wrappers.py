"""Drop-in replacements for the reference's two native extension modules, backed by the HIP kernels.

    cpp_subsampling.subsample_batch(points, batches, sampleDl=, max_p=, verbose=) -> (s_points f32 (M,3), s_len i32 (B,))
        /root/reference/src/models/backbone_kpconv/cpp_wrappers/cpp_subsampling/wrapper.cpp:62-333 (kwlist :75, return :322)
    cpp_neighbors.batch_query(queries, supports, q_batches, s_batches, radius=) -> i32 (Nq, max_count)
        /root/reference/src/models/backbone_kpconv/cpp_wrappers/cpp_neighbors/wrapper.cpp:58-238 (parse :71-75, return :214-227)

numpy in / numpy out like the originals (torch CUDA tensors are also accepted and returned); shape errors raise
RuntimeError as the originals do.  Row orders are the canonical ones documented in kpconv.py; inside
`with cpp_wrappers.reference_order():` they are the reference's own (libstdc++ unordered_map iteration order for the
subsampled rows, nanoflann visiting order + std::sort for the neighbour rows -- the parity mode of kpconv.py), and the
results equal the originals' element for element.  The switch is a field of the thread-local call context
(regtr_amd/context.py), not a module global: what one host thread selects, another does not see.
"""
import numpy as np
import torch

from . import _lib, context, ops


class _ReferenceOrder:
    """What `reference_order()` returns: a context manager.  Until round 4 `reference_order(True)` was a SETTER; a bare call of that kind is
    now a no-op, so an instance that is dropped without ever being entered says so (RuntimeWarning) instead of silently leaving the caller
    on the canonical row orders.  `set_reference_order(on)` is the setter for callers that cannot use a `with` block."""

    def __init__(self, on):
        self._on, self._ctx, self._entered = bool(on), None, False

    def __enter__(self):
        self._entered = True
        self._ctx = context.current().derive(reference_order=self._on)
        return self._ctx.__enter__()

    def __exit__(self, *exc):
        ctx, self._ctx = self._ctx, None
        return ctx.__exit__(*exc)

    def __del__(self):
        if not self._entered:
            import warnings
            warnings.warn('cpp_wrappers.reference_order(...) was called but never entered: it is a context manager (`with cpp_wrappers.'
                          'reference_order():`), the call alone changes nothing -- use cpp_wrappers.set_reference_order(on) for a setter',
                          RuntimeWarning, stacklevel=2)


def reference_order(on=True):
    """`with cpp_wrappers.reference_order():` -- both ops return the reference's own row orders (parity mode) for the calls made
    inside the block on this thread."""
    return _ReferenceOrder(on)


def set_reference_order(on):
    """Setter form (the pre-round-5 behaviour of `reference_order(True)`): switches this THREAD's base call context and returns the
    previous value."""
    return context.set_thread_default(reference_order=bool(on))['reference_order']


def _ref_order():
    return context.current().reference_order


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError('regtr_amd.cpp_wrappers needs an MI355X (HIP) device; there is no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def _to_dev(a, dtype):
    if isinstance(a, torch.Tensor):
        return a.to(device=_dev(), dtype=dtype).contiguous(), True
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=_dev(), dtype=dtype), False


def _seg(lens):
    lens = np.asarray(lens.cpu() if isinstance(lens, torch.Tensor) else lens).astype(np.int64)
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=_dev()), lens


def _canonical_rows(idx, q, s):
    """Rows of a neighbour table re-ordered to ascending (d2, index), d2 in the reference's float32 arithmetic
    ((0 + dx*dx) + dy*dy) + dz*dz (nanoflann.hpp:432-440; separate torch kernels, so nothing is contracted into an FMA)."""
    pad = s.shape[0]
    sp = torch.cat([s, torch.zeros((1, 3), dtype=s.dtype, device=s.device)])
    d = q[:, None, :] - sp[idx.long()]
    d2 = ((d[..., 0] * d[..., 0]) + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    key = (d2.view(torch.int32).long() << 32) | idx.long()
    key[idx == pad] = torch.iinfo(torch.int64).max
    return torch.gather(idx, 1, torch.argsort(key, dim=1, stable=True))


class _Subsampling:
    @staticmethod
    def subsample_batch(points, batches, features=None, classes=None, sampleDl=0.1, method='barycenters', max_p=0,
                        verbose=0):
        if features is not None or classes is not None:
            raise RuntimeError('subsample_batch: features / classes are not supported (RegTR never passes them, kpconv.py:175-181)')
        pts, is_t = _to_dev(points, torch.float32)
        if pts.dim() != 2 or pts.shape[1] != 3:
            raise RuntimeError('Wrong dimensions : points.shape is not (N, 3)')          # wrapper.cpp:127-131
        seg, lens = _seg(batches)
        if int(lens.sum()) != pts.shape[0]:
            raise RuntimeError('Wrong number of points : sum(batches) != N')
        with _lib.on_device(pts.device):
            out, out_seg = ops.grid_subsample(pts, seg, pts.shape[0], float(sampleDl), row_order=int(_ref_order()))
        oseg = out_seg.cpu().numpy()
        s_len = np.diff(oseg).astype(np.int32)
        if max_p > 0 and (s_len > max_p).any():                                           # grid_subsampling.cpp:181-204
            keep = np.concatenate([np.arange(oseg[b], oseg[b] + min(s_len[b], max_p)) for b in range(len(s_len))])
            out = out[torch.from_numpy(keep).to(out.device)]
            s_len = np.minimum(s_len, max_p).astype(np.int32)
        else:
            out = out[:int(oseg[-1])]
        if is_t:
            return out, torch.from_numpy(s_len).to(out.device)
        return out.cpu().numpy(), s_len


class _Neighbors:
    @staticmethod
    def batch_query(queries, supports, q_batches, s_batches, radius=0.1):
        q, is_t = _to_dev(queries, torch.float32)
        s, _ = _to_dev(supports, torch.float32)
        if q.dim() != 2 or q.shape[1] != 3 or s.dim() != 2 or s.shape[1] != 3:
            raise RuntimeError('Wrong dimensions : queries / supports shape is not (N, 3)')   # wrapper.cpp:127-149
        qseg, qlens = _seg(q_batches)
        sseg, slens = _seg(s_batches)
        if len(qlens) != len(slens):
            raise RuntimeError('Wrong number of batch elements: different for queries and supports')   # :165-171
        if int(qlens.sum()) != q.shape[0] or int(slens.sum()) != s.shape[0]:
            raise RuntimeError('Wrong number of points : sum(batches) != N')
        with _lib.on_device(q.device):
            if _ref_order():
                tree = ops.KdTree(s, sseg, s.shape[0])
                _, width = tree.query(q, qseg, q.shape[0], float(radius), 1)          # pass 1: the row width (max in-ball count)
                idx, _ = tree.query(q, qseg, q.shape[0], float(radius), max(width, 1), list_cap=max(width, 16))
            else:
                grid = ops.CellGrid(s, sseg, s.shape[0], float(radius))
                K = 64
                while True:
                    idx, cnt, mx = grid.query(q, qseg, q.shape[0], K, want_count=True)
                    width = int(mx.item())
                    if width <= K or K >= 448:
                        break
                    K = min(448, max(2 * K, width))
                if width > K:
                    # more supports in one ball than the wavefront kernel's LDS list holds (448): the KD-tree kernel has no
                    # limit; its rows (reference order) are re-sorted to the canonical (d2, index) order.  Rare (the
                    # reference itself has no limit, neighbors.cpp:290-293) and off the model's hot path.
                    tree = ops.KdTree(s, sseg, s.shape[0])
                    idx, _ = tree.query(q, qseg, q.shape[0], float(radius), width, list_cap=width)
                    idx = _canonical_rows(idx[:q.shape[0]], q, s)
        if width == 0:
            raise RuntimeError('Error converting output: no neighbour found')           # wrapper.cpp:201-205
        idx = idx[:q.shape[0], :width].contiguous()
        return idx if is_t else idx.cpu().numpy()


grid_subsampling = _Subsampling      # `from ...cpp_subsampling import grid_subsampling as cpp_subsampling` (kpconv.py:14)
radius_neighbors = _Neighbors        # `from ...cpp_neighbors import radius_neighbors as cpp_neighbors`     (kpconv.py:15)
