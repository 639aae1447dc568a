"""Ground-truth overlap on the GPU (SURVEY.md section 8 f4 -- the training / validation side of the hot path's callers).

    compute_overlap(src, tgt, search_voxel_size)   /root/reference/src/utils/pointcloud.py:8-65
    compute_overlaps(batch)                        /root/reference/src/models/backbone_kpconv/kpconv.py:540-566

Same signatures and return values as the reference functions; the radius matching runs on the cell-grid kernels of
csrc/preprocess.hip (float64 distances over float32 coordinates, like open3d's KDTreeFlann on the widened points) and the
pyramid pooling on regtr_overlap_avgpool.  No CPU fallback.
"""
import numpy as np
import torch

from . import _lib, ops
from ._lib import check, iptr, ptr, stream


def nearest_in_radius(q, q_seg, s, s_seg, radius):
    """Per query: index (into the stacked supports) of the nearest support of the same cloud slot with d2 < radius^2, else -1."""
    L = _lib.lib()
    r32 = float(np.float32(radius))
    grid = ops.CellGrid(s, s_seg, s.shape[0], r32)
    out = torch.empty(max(q.shape[0], 1), dtype=torch.int32, device=q.device)
    check(L.regtr_nearest_in_radius(ptr(q), iptr(q_seg), q.shape[0], iptr(s_seg), s.shape[0], s_seg.numel() - 1, float(radius),
                                    r32, _lib.bptr(grid.ws), grid.nbytes, iptr(out), stream()), 'regtr_nearest_in_radius')
    return out[:q.shape[0]]


def _dev_cloud(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def compute_overlap(src, tgt, search_voxel_size, device=None):
    """utils/pointcloud.py:8-65.  src, tgt: (N, 3) arrays (numpy or torch).  Returns (has_corr_src (Ns,) bool, has_corr_tgt (Nt,)
    bool, src_tgt_corr (2, M) int64) as numpy arrays -- including the reference's `src_corr > 0` quirk (:58-59), which drops
    mutual matches onto target index 0."""
    if device is None:
        device = src.device if isinstance(src, torch.Tensor) and src.is_cuda else torch.device('cuda', torch.cuda.current_device())
    with _lib.on_device(device):
        s, t = _dev_cloud(src, device), _dev_cloud(tgt, device)
        seg = lambda n: torch.tensor([0, n], dtype=torch.int32, device=device)
        tgt_corr = nearest_in_radius(t, seg(len(t)), s, seg(len(s)), search_voxel_size).cpu().numpy().astype(np.int64)   # :44-49
        src_corr = nearest_in_radius(s, seg(len(s)), t, seg(len(t)), search_voxel_size).cpu().numpy().astype(np.int64)   # :50-55
    src_corr_is_mutual = np.logical_and(tgt_corr[src_corr] == np.arange(len(src_corr)), src_corr > 0)                    # :58-59
    src_tgt_corr = np.stack([np.nonzero(src_corr_is_mutual)[0], src_corr[src_corr_is_mutual]])
    return src_corr >= 0, tgt_corr >= 0, src_tgt_corr


def compute_overlaps(batch):
    """kpconv.py:540-566: ground-truth overlap of every pyramid level by average-pooling the level below through the
    `pools` tables of batch['kpconv_meta'].  batch['src_overlap'] / ['tgt_overlap']: lists (B) of per-point masks."""
    meta = batch['kpconv_meta']
    overlaps = list(batch['src_overlap']) + list(batch['tgt_overlap'])
    dev = meta['points'][0].device
    with _lib.on_device(dev):
        cur = torch.cat([o.to(dev) for o in overlaps], dim=0).to(torch.float32).contiguous()
        pyr = {'pyr_0': cur}
        for p in range(1, len(meta['points'])):
            pools = meta['_pools_i32'][p - 1] if '_pools_i32' in meta else meta['pools'][p - 1].to(torch.int32).contiguous()
            nq, ld = pools.shape
            H = meta['_pool_width'][p - 1] if '_pool_width' in meta else ld
            out = torch.empty(nq, dtype=torch.float32, device=dev)
            check(_lib.lib().regtr_overlap_avgpool(ptr(cur), cur.shape[0], iptr(pools), ld, nq, int(H), ptr(out), stream()),
                  'regtr_overlap_avgpool')
            pyr[f'pyr_{p}'] = cur = out
    return pyr
