"""KPConv preprocessing pyramid and encoder on the HIP kernels.

Host-side mirror of /root/reference/src/models/backbone_kpconv/kpconv.py (Preprocessor :291-414 / PreprocessorGPU
:417-537, KPFEncoder :22-88) and kpconv_blocks.py (SimpleBlock :590-646, ResnetBottleneckBlock :649-741, UnaryBlock
:533-567, KPConv :175-414): same constructor arguments, same parameter names, same `kpconv_meta` dictionary.  All
arithmetic is in libregtr_hip.so.

Default semantics are those of the reference's CPU ops (cpp_wrappers; the only path whose outputs can be generated and pinned in a
container without MinkowskiEngine / pytorch3d): voxel key floor((p - origin) / dl), neighbours distance-sorted then truncated to
neighborhood_limits.  Documented differences, none of which changes a downstream value beyond float rounding and ties: neighbour
tables always have K = neighborhood_limits[l] columns like the reference's GPU path (kpconv.py:276-283; its CPU path emits
min(max_count, K), kpconv.py:255-258 -- the extra columns are shadow indices), exact-distance ties are ordered by support index,
subsampled rows are in first-appearance order, index tensors are int32 unless cfg.kpconv_meta_int64 is set, and the unused
`upsamples` tables (kpconv.py:503-504; RegTR has no decoder) are left empty.

cfg.kpconv_ref_row_order = True selects the PARITY MODE: every implementation-defined choice of the reference's CPU
ops is reproduced on the GPU -- subsampled rows in libstdc++ unordered_map iteration order, neighbour rows in nanoflann
visiting order passed through std::sort (which decides WHICH equidistant supports a truncated row keeps), tables
min(max_count, K) wide (a full row then has no zero shadow row in max_pool) -- so `kpconv_meta` equals the reference
`Preprocessor`'s output element for element and the network outputs can be compared with the reference's own at 1e-4
(tests/test_gpu_model.py).  It costs extra host synchronisations and serial kernels; the default mode stays the fast one.

The reference MODEL instantiates PreprocessorGPU (regtr.py:29), whose two ops follow different rules from the CPU ones -- both are
selectable, together they are that class's semantics (test.py --preprocessor gpu sets both):
 * cfg.kpconv_neighbor_order = 'index': pytorch3d ball_query keeps the FIRST K supports of a ball in index order
   (kpconv.py:261-288), where the CPU Preprocessor keeps the K nearest.  The two differ on rows whose ball holds more than
   neighborhood_limits supports (15 % of level-0 rows on 3DMatch).  Rows come out ascending by support index.
 * cfg.kpconv_voxel_key = 'floor': MinkowskiEngine quantises points / sampleDl, i.e. the voxel of p is floor(p / dl) with NO origin
   shift (kpconv.py:230-239), where the CPU op takes floor((p - origin) / dl) with origin = floor(min corner / dl) * dl
   (grid_subsampling.cpp:25-31,53-55).  These are DIFFERENT voxel sets on real data: 3DMatch fragments sit on a lattice that puts
   whole planes of points exactly on voxel faces, where (p - origin) / dl and p / dl round to different sides -- the red-kitchen
   pair subsamples to 9 977 level-1 points under the CPU rule and 10 088 under this one.  'floor_rcp' evaluates p * (1 / dl) with a
   float32 reciprocal, which is what torch's CUDA kernel computes for a division by a host scalar (10 037 points on the same pair).
Both run at the default mode's speed.  pytorch3d / MinkowskiEngine cannot be installed here, so these modes are held to
restatements of the call sites' documented rules (oracle/regtr_ref.py: ball_query_first_k; oracle/regtr_oracle.cpp:
oracle_grid_subsample_keyed) -- parity UNPINNED for them; MinkowskiEngine's row and summation orders are unspecified in the reference
itself (kpconv.py:216-217), first-appearance order and in-order float32 sums are used.
"""
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import _lib, context, ops
from .kernel_points import load_kernels


def param_fingerprint(module):
    """Cheap identity of every parameter under `module`: (storage address, version) of each, fetched through a cached list of (owner module,
    name) pairs -- `module.parameters()` walks the module tree through four generator layers (0.3 ms for the cross-encoder's 108 parameters,
    tools/host_profile.py), a dict lookup per parameter costs 50 us for all of them, and unlike a cached list of Parameter OBJECTS it still
    sees a parameter that was replaced (`layer.weight = nn.Parameter(...)`).  The (owner, name) list itself only changes when submodules are
    added or removed, which none of these containers does after construction."""
    slots = module.__dict__.get('_regtr_param_slots')
    if slots is None:
        slots = [(m, n) for m in module.modules() for n in m._parameters]
        module.__dict__['_regtr_param_slots'] = slots
    fp = []
    for m, n in slots:
        prm = m._parameters[n]
        if prm is not None:
            fp.append(prm.data_ptr()); fp.append(prm._version)
    return tuple(fp)


def _prepared(cache, key, param, fn):
    """Weights re-laid-out for the kernels, cached until the parameter changes."""
    ent = cache.get(key)
    tag = (param.data_ptr(), param._version, param.device)
    if ent is None or ent[0] != tag:
        with torch.no_grad():
            ent = (tag, fn(param.detach()))
        cache[key] = ent
    return ent[1]


CAPACITY_MIN_POINTS = 262144      # below this many input points every level is sized at N_0 (nothing to save, no rebuild risk)


class Preprocessor(nn.Module):
    """Computes the metadata used for KPConv (kpconv.py:291 / :417).  One host synchronisation per call: the whole
    pyramid is enqueued against capacity-sized buffers, then the per-level segment offsets are read back once."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self._seg_stage = None      # pinned staging buffer of the cloud offsets (enqueue) and the event behind its last copy
        self._seg_copied = None
        # Level sizes are data dependent and known only on the device while the pyramid is enqueued, so every level's buffers are sized
        # from a CAPACITY: level l holds at most capacity[l] x N_0 points.  Each level doubles the voxel size -- a surface loses ~3/4 of
        # its points per level (3DMatch: 27 % / 7 % / 2 % of N_0), a curve 1/2 -- so the default halves the capacity per level, which
        # sizes the seven K-column neighbour tables of a 64-pair forward at 1.1 GB instead of 2.7 GB.  A level that FILLS its capacity
        # saturates on the device (regtr_grid_subsample_ordered's out_cap), is detected at the one size read-back, and the pyramid is
        # rebuilt at full capacity (N_{l+1} <= N_l always holds) -- from then on this preprocessor stays at full capacity.
        self.level_capacity = list(cfg.get('kpconv_level_capacity', [1.0, 0.5, 0.25, 0.125, 0.0625, 0.03125]))
        self.capacity_overflows = 0

    def forward(self, pts: List[torch.Tensor]):
        meta = self.finish(self.enqueue(pts))
        if meta is None:                      # a level filled its capacity: once more, at full capacity
            meta = self.finish(self.enqueue(pts))
        return meta

    def enqueue(self, pts: List[torch.Tensor], level0_event=None):
        """Enqueues the whole pyramid on the current stream without touching the host (default order only; the parity mode's KD-tree
        reads its row widths back).  level0_event: recorded as soon as level 0's conv table is complete -- everything the level-0
        blocks need (RegTR.forward starts them on its main stream while the rest of the pyramid is still being built on this one).
        -> state for finish() / level0_meta()."""
        cfg = self.cfg
        limits = cfg.neighborhood_limits
        device = pts[0].device
        lens0 = [int(p.shape[0]) for p in pts]
        n0 = sum(lens0)
        points = torch.cat([p.to(torch.float32) for p in pts], dim=0).contiguous()
        # cloud offsets: through a pinned staging buffer and an asynchronous copy (torch.tensor(list, device=...) is a blocking pageable
        # copy, ~0.1 ms of a 2.7 ms one-pair forward).  The buffer is rewritten only by the next enqueue(), after finish() has waited
        # for this pyramid (and with it for the copy).
        stage = self._seg_stage
        if self._seg_copied is not None:
            self._seg_copied.synchronize()          # (already complete in every present caller: they finish() before the next enqueue())
        if stage is None or stage.numel() != len(lens0) + 1:
            stage = self._seg_stage = torch.empty(len(lens0) + 1, dtype=torch.int32).pin_memory()
        stage_np = stage.numpy()
        stage_np[0] = 0
        np.cumsum(lens0, out=stage_np[1:])
        seg = stage.to(device, non_blocking=True)
        self._seg_copied = torch.cuda.Event()
        self._seg_copied.record()
        # parity mode: the reference CPU ops' implementation-defined row orders and table widths (see module docstring)
        ref_order = bool(cfg.get('kpconv_ref_row_order', False))
        nb_order = {'nearest': 0, 'index': 1}[cfg.get('kpconv_neighbor_order', 'nearest')]
        key_name = cfg.get('kpconv_voxel_key', 'origin')
        if key_name not in ops.VOXEL_KEY_MODES:
            raise ValueError(f'kpconv_voxel_key {key_name!r}: choose origin (CPU Preprocessor), floor or floor_rcp (PreprocessorGPU)')
        key_mode = ops.VOXEL_KEY_MODES[key_name]
        if ref_order and (nb_order or key_mode):
            raise ValueError('kpconv_ref_row_order reproduces the CPU Preprocessor; kpconv_neighbor_order = index / kpconv_voxel_key = floor are the GPU one')

        r_normal = cfg.first_subsampling_dl * cfg.conv_radius                 # kpconv.py:315
        layer_blocks, layer = [], 0
        lv_points, lv_seg, lv_conv, lv_pool, lv_width, lv_cap = [], [], [], [], [], []
        cap = n0                                                              # N_{l+1} <= N_l <= N_0
        arch = cfg.architecture
        for block_i, block in enumerate(arch):                               # kpconv.py:328-404
            if 'global' in block or 'upsample' in block:
                break
            if not ('pool' in block or 'strided' in block):
                layer_blocks += [block]
                if block_i < len(arch) - 1 and not ('upsample' in arch[block_i + 1]):
                    continue
            if any('deformable' in b for b in layer_blocks) or 'deformable' in block:
                raise NotImplementedError('deformable KPConv is outside the RegTR inference path')
            K = limits[layer]
            strided = 'pool' in block or 'strided' in block
            ratio = self.level_capacity[layer + 1] if layer + 1 < len(self.level_capacity) else self.level_capacity[-1]
            # (parity mode replays containers over whole clouds; a pair or two per forward: the tables are small, full capacity)
            cap_next = cap if (ref_order or n0 < CAPACITY_MIN_POINTS) else min(cap, max(int(n0 * ratio) + 64, 1))
            dl = 2 * r_normal / cfg.conv_radius                                          # :363
            conv_i = pool_p = pool_seg = pool_i = None
            conv_w = pool_w = K
            if not ref_order:
                grid = ops.CellGrid(points, seg, cap, r_normal)
                if layer_blocks:
                    conv_i = grid.query(points, seg, cap, K, order=nb_order)             # :349-351
                if layer == 0 and level0_event is not None:
                    level0_event.record()
                if strided:
                    pool_p, pool_seg = ops.grid_subsample(points, seg, cap, dl, key_mode=key_mode, out_cap=cap_next)          # :366 / :213-240
                    pool_i = grid.query(pool_p, pool_seg, cap_next, K, order=nb_order)   # :376
            else:
                tree = ops.KdTree(points, seg, cap)
                if layer_blocks:
                    conv_i, conv_w = tree.query(points, seg, cap, r_normal, K)
                if strided:
                    pool_p, pool_seg = ops.grid_subsample(points, seg, cap, dl, row_order=1)
                    pool_i, pool_w = tree.query(pool_p, pool_seg, cap, r_normal, K)
            lv_points.append(points); lv_seg.append(seg); lv_conv.append(conv_i); lv_pool.append(pool_i); lv_cap.append(cap)
            lv_width.append((min(conv_w, K), min(pool_w, K)))                            # kpconv.py:255-258
            points, seg = pool_p, pool_seg
            if strided:
                cap = cap_next
            r_normal *= 2
            layer += 1
            layer_blocks = []
        # the level sizes travel to pinned host memory behind the last kernel; `done` is all a consumer (host or stream) waits for
        seg_pin = torch.empty((len(lv_seg), len(lens0) + 1), dtype=torch.int32, pin_memory=True)
        seg_pin.copy_(torch.stack(lv_seg), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return {'lens0': lens0, 'device': device, 'ref_order': ref_order, 'lv_points': lv_points, 'lv_seg': lv_seg, 'lv_conv': lv_conv,
                'lv_pool': lv_pool, 'lv_width': lv_width, 'lv_cap': lv_cap, 'seg_pin': seg_pin, 'done': done}

    @staticmethod
    def level0_meta(state):
        """What the level-0, non-strided blocks read from kpconv_meta (kpconv.py: _LevelView) -- all sizes are the inputs' own, so it
        exists before the pyramid's level sizes have been read back."""
        return {'points': [state['lv_points'][0]], '_neighbors_i32': [state['lv_conv'][0]], '_seg_off': [state['lv_seg'][0]],
                '_lens_host': [state['lens0']]}

    def finish(self, state):
        """The one host round trip (level sizes) and the kpconv_meta dictionary.  Runs on the CONSUMER's stream: it waits for the
        pyramid's `done` event only, so a pyramid enqueued on another stream (RegTR's side stream, a prefetch for the next batch)
        does not drag that stream's later work in."""
        cfg = self.cfg
        device, ref_order = state['device'], state['ref_order']
        lv_points, lv_seg, lv_conv, lv_pool, lv_width = (state[k] for k in ('lv_points', 'lv_seg', 'lv_conv', 'lv_pool', 'lv_width'))
        state['done'].synchronize()                                                      # host: the sizes have landed
        torch.cuda.current_stream().wait_event(state['done'])                            # stream: the tables are complete
        seg_host = state['seg_pin'].numpy()                                              # (levels, n_clouds + 1)
        caps = state['lv_cap']
        full = [l for l in range(1, len(caps)) if caps[l] < caps[l - 1] and int(seg_host[l, -1]) >= caps[l]]
        if full:        # a subsampled level filled its buffer (saturated on the device): the caller rebuilds at full capacity
            self.capacity_overflows += 1
            self.level_capacity = [1.0] * len(self.level_capacity)
            import logging
            logging.getLogger('regtr_amd').warning(
                'KPConv pyramid: level(s) %s filled their capacity (%s of N_0 = %d points); rebuilding at full capacity (cfg.kpconv_level_capacity)',
                full, [round(caps[l] / max(caps[0], 1), 3) for l in full], caps[0])
            return None
        data = {'points': [], 'neighbors': [], 'pools': [], 'upsamples': [], 'stack_lengths': [],
                '_seg_off': lv_seg, '_lens_host': [], '_neighbors_i32': [], '_pools_i32': [], '_pool_width': []}
        want64 = bool(cfg.get('kpconv_meta_int64', False))
        # stack_lengths of every level in one upload (they are already on the host) instead of two small kernels per level
        stack_lengths = state.get('stack_lengths')
        if stack_lengths is None:
            stack_lengths = torch.from_numpy(np.diff(seg_host, axis=1).astype(np.int64)).to(device)
        for l in range(len(lv_points)):
            n_l = int(seg_host[l, -1])
            n_next = int(seg_host[l + 1, -1]) if l + 1 < len(lv_points) else 0
            lens = np.diff(seg_host[l]).astype(np.int64)
            data['_lens_host'].append(lens.tolist())
            data['points'].append(lv_points[l][:n_l])
            conv = lv_conv[l][:n_l] if lv_conv[l] is not None else torch.zeros((0, 1), dtype=torch.int32, device=device)
            pool = lv_pool[l][:n_next] if lv_pool[l] is not None else torch.zeros((0, 1), dtype=torch.int32, device=device)
            data['_neighbors_i32'].append(conv)
            data['_pools_i32'].append(pool)
            data['_pool_width'].append(lv_width[l][1])
            if ref_order:        # the reference's table shapes: min(max in-ball count, K) columns (kpconv.py:255-258)
                conv = conv[:, :lv_width[l][0]].contiguous() if lv_conv[l] is not None else conv
                pool = pool[:, :lv_width[l][1]].contiguous() if lv_pool[l] is not None else pool
            data['neighbors'].append(conv.long() if want64 else conv)
            data['pools'].append(pool.long() if want64 else pool)
            data['upsamples'].append(torch.zeros((0, 1), dtype=torch.int64, device=device))
            data['stack_lengths'].append(stack_lengths[l])
        return data


PreprocessorGPU = Preprocessor   # the reference instantiates PreprocessorGPU (regtr.py:29); same contract here


# ----------------------------------------------------------------------------------------------------------------
# blocks (parameter containers + HIP forward)
# ----------------------------------------------------------------------------------------------------------------
class KPConv(nn.Module):
    """Parameters of kpconv_blocks.py:175-267 (non-deformable)."""

    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius, fixed_kernel_points='center',
                 KP_influence='linear', aggregation_mode='sum', deformable=False, modulated=False):
        super().__init__()
        if deformable or KP_influence != 'linear' or aggregation_mode != 'sum':
            raise NotImplementedError('only rigid KPConv with linear influence and sum aggregation is implemented '
                                      '(the only mode either reference config uses)')
        self.K, self.p_dim = kernel_size, p_dim
        self.in_channels, self.out_channels = in_channels, out_channels
        self.radius, self.KP_extent = radius, KP_extent
        self.weights = nn.Parameter(torch.zeros((kernel_size, in_channels, out_channels), dtype=torch.float32))
        nn.init.kaiming_uniform_(self.weights, a=5 ** 0.5)                      # kpconv_blocks.py:248-249
        kp = load_kernels(radius, kernel_size, dimension=p_dim, fixed=fixed_kernel_points)
        self.kernel_points = nn.Parameter(torch.tensor(kp, dtype=torch.float32), requires_grad=False)   # :266
        self._cache = {}

    def forward(self, q_pts, s_pts, neighb_inds, x, x_stats=None, s_seg_off=None, q_seg_off=None, want_stats=None, xyzf=None):
        w = _prepared(self._cache, 'w', self.weights,
                      lambda p: ops.SplitWeight(p.view(self.K * self.in_channels, self.out_channels), 'kn'))
        return ops.kpconv(q_pts, s_pts, neighb_inds, x, w, self.kernel_points.detach(), self.KP_extent,
                          x_stats=x_stats, s_seg_off=s_seg_off, q_seg_off=q_seg_off, want_stats=want_stats, xyzf=xyzf)


class UnaryBlock(nn.Module):
    """Linear(no bias) -> per-cloud InstanceNorm -> LeakyReLU(0.1) (kpconv_blocks.py:533-567)."""

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super().__init__()
        if not use_bn:
            raise NotImplementedError('use_batch_norm=False (bias instead of InstanceNorm) is not implemented')
        self.in_dim, self.out_dim, self.no_relu = in_dim, out_dim, no_relu
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self._cache = {}

    def linear(self, x, seg_off, max_len, a_stats=None, a_seg_off=None):
        """The Linear and the InstanceNorm statistics of its output (from the GEMM epilogue); the normalisation itself
        (+LeakyReLU) is folded into whichever kernel consumes the result.  -> (y, stats)"""
        wt = _prepared(self._cache, 'w', self.mlp.weight, lambda w: ops.SplitWeight(w, 'nk'))
        return ops.gemm(x, wt, a_stats=a_stats, a_seg_off=a_seg_off, want_stats=(seg_off, max_len))

    def forward(self, x, seg_off, max_len):
        y, st = self.linear(x, seg_off, max_len)
        return ops.instnorm_apply(y, seg_off, max_len, st, lrelu=not self.no_relu, out=y)


class _LevelView:
    """What a block needs from kpconv_meta for its (possibly strided) convolution."""

    def __init__(self, meta, layer, strided):
        self.s_pts = meta['points'][layer]
        self.q_pts = meta['points'][layer + 1] if strided else self.s_pts
        self.inds = meta['_pools_i32'][layer] if strided else meta['_neighbors_i32'][layer]
        self.pool_width = meta['_pool_width'][layer] if strided else None
        self.seg_pre = meta['_seg_off'][layer]
        self.max_pre = max(meta['_lens_host'][layer])
        self.seg_post = meta['_seg_off'][layer + 1] if strided else self.seg_pre
        self.max_post = max(meta['_lens_host'][layer + 1]) if strided else self.max_pre
        self.small = meta['points'][0].shape[0] < ops.SMALL_REGIME_ROWS       # the small-batch regime (ops.SMALL_REGIME_ROWS)


class SimpleBlock(nn.Module):
    """KPConv -> InstanceNorm -> LeakyReLU (kpconv_blocks.py:590-646)."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius          # :603
        self.block_name, self.layer_ind = block_name, layer_ind
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim // 2, current_extent,
                             radius, fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                             aggregation_mode=config.aggregation_mode, deformable='deform' in block_name,
                             modulated=config.modulated)
        if not config.use_batch_norm:
            raise NotImplementedError('use_batch_norm=False is not implemented')

    def forward(self, x, meta):
        v = _LevelView(meta, self.layer_ind, 'strided' in self.block_name)
        # one input feature per point (RegTR's ones): (x, y, z, feature) records, one 16-byte load per neighbour in the gather
        xyzf = torch.cat((v.s_pts, x), dim=1) if (x.shape[1] == 1 and ops.prenorm_gather and x.shape[0] >= ops.PRENORM_MIN_ROWS and not v.small) else None
        kp = self.KPConv
        if x.shape[1] == 1 and ops.first_block_ok(v.q_pts.shape[0], 1, kp.K, kp.out_channels):
            # contraction + InstanceNorm + LReLU in one pass over the gather's 16-float rows (csrc/block_tail.hip)
            w16 = _prepared(kp._cache, 'w16', kp.weights,
                            lambda p: torch.cat((p.view(kp.K, kp.out_channels), p.new_zeros(1, kp.out_channels)), 0).contiguous())
            return ops.kpconv_norm_lrelu(v.q_pts, v.s_pts, v.inds, x, w16, kp.kernel_points.detach(), kp.KP_extent,
                                         v.seg_post, v.max_post, xyzf=xyzf)
        y, st = self.KPConv(v.q_pts, v.s_pts, v.inds, x, want_stats=(v.seg_post, v.max_post), xyzf=xyzf)
        return ops.instnorm_apply(y, v.seg_post, v.max_post, st, lrelu=True, out=y)


class ResnetBottleneckBlock(nn.Module):
    """unary1 -> KPConv -> IN -> LReLU -> unary2 ; shortcut [max_pool][unary] ; LReLU(sum)
    (kpconv_blocks.py:649-741)."""

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super().__init__()
        current_extent = radius * config.KP_extent / config.conv_radius          # :662
        bn, mom = config.use_batch_norm, config.batch_norm_momentum
        self.block_name, self.layer_ind = block_name, layer_ind
        self.unary1 = UnaryBlock(in_dim, out_dim // 4, bn, mom) if in_dim != out_dim // 4 else nn.Identity()
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, out_dim // 4, out_dim // 4, current_extent,
                             radius, fixed_kernel_points=config.fixed_kernel_points, KP_influence=config.KP_influence,
                             aggregation_mode=config.aggregation_mode, deformable='deform' in block_name,
                             modulated=config.modulated)
        self.unary2 = UnaryBlock(out_dim // 4, out_dim, bn, mom, no_relu=True)
        self.unary_shortcut = UnaryBlock(in_dim, out_dim, bn, mom, no_relu=True) if in_dim != out_dim else nn.Identity()

    def forward(self, features, meta):
        strided = 'strided' in self.block_name
        v = _LevelView(meta, self.layer_ind, strided)
        # unary1 = Linear -> IN -> LReLU (:722).  The IN + LReLU tail runs once per support row, in place, and the same pass packs
        # (x, y, z, "feature sum > 0" flag of KPConv's normaliser, :409-410) into 16-byte records: the gather then issues one load per
        # neighbour instead of four scattered dwords (it is bound by the texture-address path) and neither folds nor sums rows.
        # (ops.prenorm_gather = False: everything folded into the gather.)
        xyzf = None
        if isinstance(self.unary1, UnaryBlock):
            x, x_st = self.unary1.linear(features, v.seg_pre, v.max_pre)
            if ops.prenorm_gather and x.shape[1] <= 256 and x.shape[0] >= ops.PRENORM_MIN_ROWS and not v.small:   # (small batches: launch-bound, fold)
                xyzf = torch.empty((x.shape[0], 4), dtype=torch.float32, device=x.device)
                ops.instnorm_apply(x, v.seg_pre, v.max_pre, x_st, lrelu=True, out=x, row_xyz=v.s_pts, row_positive=xyzf)
                x_st = None
        else:
            x, x_st = features, None
        x, st = self.KPConv(v.q_pts, v.s_pts, v.inds, x, x_stats=x_st, s_seg_off=v.seg_pre, q_seg_off=v.seg_post,
                            want_stats=(v.seg_post, v.max_post), xyzf=xyzf)                                   # :726
        # Linear shortcut on the same rows (not strided): the whole tail -- unary2, shortcut, both InstanceNorms, sum, LReLU -- in one
        # pass over the two narrow inputs, statistics from their second moments (csrc/block_tail.hip)
        if not strided and isinstance(self.unary_shortcut, UnaryBlock):
            w2 = _prepared(self.unary2._cache, 'w', self.unary2.mlp.weight, lambda w: ops.SplitWeight(w, 'nk'))
            ws = _prepared(self.unary_shortcut._cache, 'w', self.unary_shortcut.mlp.weight, lambda w: ops.SplitWeight(w, 'nk'))
            if ops.block_tail_ok(x, st, features, w2, ws):
                return ops.block_tail(x, st, features, w2, ws, v.seg_post, v.max_post)
        shortcut = ops.maxpool(features, v.inds, v.pool_width) if strided else features                                     # :734-737
        sc_st = None
        if isinstance(self.unary_shortcut, UnaryBlock):
            shortcut, sc_st = self.unary_shortcut.linear(shortcut, v.seg_post, v.max_post)
        # IN + LReLU of the convolution output (:727): folded into unary2's GEMM A-operand load (:730) where the one-shot strip kernel takes
        # the fold (K <= 64); for the deeper levels (K >= 128) the fold would route the product to the tiled kernel (A staged through
        # registers, two barriers per k-tile: 246 us against 118 us for the same shape on the row-strip kernel at level 3), so the narrow
        # conv output is normalised in place first (one pass over [M, K]: 20-35 us) and the row-strip kernel multiplies it
        if ops.preapply_unary2 and st is not None and x.shape[0] >= ops.PREAPPLY_MIN_ROWS and (ops.preapply_unary2 >= 2 or x.shape[1] > 64):
            ops.instnorm_apply(x, v.seg_post, v.max_post, st, lrelu=True, out=x)
            st = None
        y, y_st = self.unary2.linear(x, v.seg_post, v.max_post, a_stats=st, a_seg_off=v.seg_post if st is not None else None)
        # LeakyReLU( IN(unary2) + [IN](shortcut) ) in one pass                                                :741
        return ops.instnorm_apply(y, v.seg_post, v.max_post, y_st, residual=shortcut, res_stats=sc_st, lrelu=True, out=y)


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    """kpconv_blocks.py:429-471, restricted to the block types of the RegTR encoders."""
    if block_name in ('simple', 'simple_strided'):
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ('resnetb', 'resnetb_strided'):
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    raise NotImplementedError(f'block "{block_name}" is outside the RegTR inference path')


class KPFEncoder(nn.Module):
    """kpconv.py:22-88."""

    def __init__(self, config, d_bottle, increase_channel_when_downsample=True):
        super().__init__()
        octave = 0
        r = config.first_subsampling_dl * config.conv_radius
        in_dim, out_dim = config.in_feats_dim, config.first_feats_dim
        self.encoder_blocks = nn.ModuleList()
        self.encoder_skip_dims, self.encoder_skips = [], []
        block = None
        for block_i, block in enumerate(config.architecture):
            if any(t in block for t in ('pool', 'strided', 'upsample', 'global')):
                self.encoder_skips.append(block_i)
                self.encoder_skip_dims.append(in_dim)
            if 'upsample' in block:
                break
            self.encoder_blocks.append(block_decider(block, r, in_dim, out_dim, octave, config))
            in_dim = out_dim // 2 if 'simple' in block else out_dim
            if 'pool' in block or 'strided' in block:
                octave += 1
                r *= 2
                if increase_channel_when_downsample:
                    out_dim *= 2
        if 'upsample' not in block:
            self.encoder_skips.append(block_i)
            self.encoder_skip_dims.append(in_dim)

    def forward(self, x, batch, start=0, stop=None, skip_x=None):
        """Blocks [start, stop) (all by default); skip_x carries the skip list across a split call."""
        skip_x = [] if skip_x is None else skip_x
        stop_i = len(self.encoder_blocks) if stop is None else min(stop, len(self.encoder_blocks))
        if start < stop_i and self._one_call_ok(x, batch):
            y = self._forward_one_call(x, batch, start, stop_i)
            if y is not None:
                return y, skip_x            # (the skip list feeds a decoder RegTR does not have, kpconv.py:93-94: not collected on this path)
        for block_i, block_op in enumerate(self.encoder_blocks):
            if block_i < start or (stop is not None and block_i >= stop):
                continue
            if block_i in self.encoder_skips:
                skip_x.append(x)
            x = block_op(x, batch)
        return x, skip_x

    # ---- blocks [start, stop) through ONE C call (regtr_encoder_fwd, csrc/encoder.hip): the same launches in the same order, sequenced in C.
    # Small batches only (a pair or two per forward: the reference's own mode), where the host is the bound of an op-by-op forward.
    def _one_call_ok(self, x, meta):
        ctx = context.current()
        return bool(ops.use_one_call_encoder and ctx.gather_records is None and ctx.gemm_records is None and ctx.f16_range_log is None
                    and not ops.force_f32_gemm and not ops.force_x3_gemm and ops.use_tile_info and ops.preapply_unary2 == 1
                    and meta['points'][0].shape[0] < min(ops.SMALL_REGIME_ROWS, ops.STREAM_MIN_ROWS) and x.dim() == 2 and x.is_contiguous() and x.data_ptr() % 16 == 0
                    and x.shape[0] > 0)

    def _block_table(self):
        """ctypes array of regtr_encoder_block_t for the blocks, rebuilt when a parameter changes (storage address or version: a cheap
        fingerprint per forward); the tensors behind the pointers are kept alive by the blocks' weight caches."""
        fp = param_fingerprint(self)
        if getattr(self, '_table', None) is not None and self._table[0] == fp:
            return self._table[1]
        rows = []
        for blk in self.encoder_blocks:
            kp = blk.KPConv
            ws = [None, None, None, None]
            ws[1] = _prepared(kp._cache, 'w', kp.weights, lambda p, kp=kp: ops.SplitWeight(p.view(kp.K * kp.in_channels, kp.out_channels), 'kn'))
            kind = 0
            if isinstance(blk, ResnetBottleneckBlock):
                kind = 1
                for i, u in ((0, blk.unary1), (2, blk.unary2), (3, blk.unary_shortcut)):
                    if isinstance(u, UnaryBlock):
                        ws[i] = _prepared(u._cache, 'w', u.mlp.weight, lambda w: ops.SplitWeight(w, 'nk'))
            rows.append((blk, kind, ws))
        arr = (_lib.EncoderBlock * len(rows))()
        keep = []
        for e, (blk, kind, ws) in zip(arr, rows):
            kp = blk.KPConv
            e.kind, e.strided, e.layer = kind, int('strided' in blk.block_name), int(blk.layer_ind)
            e.n_kp, e.extent = int(kp.K), float(kp.KP_extent)
            e.kernel_points = _lib.ptr(kp.kernel_points.detach())
            for name, w in zip(('unary1', 'conv', 'unary2', 'shortcut'), ws):
                f = getattr(e, name)
                if w is None:
                    f.kn = f.planes = f.planes16 = None
                    f.N = f.K = 0
                    continue
                f.kn = _lib.ptr(w.kn)
                f.planes = _lib.bptr(w.planes) if w.planes is not None else None
                f.planes16 = _lib.bptr(w.planes16) if (w.planes is not None and w.f16_ok) else None
                f.N, f.K = int(w.N), int(w.K)
                keep.append(w)
        self._table = (fp, arr, keep)
        return arr

    def _forward_one_call(self, x, meta, start, stop):
        L = _lib.lib()
        n_levels = len(meta['points'])
        n_clouds = meta['_seg_off'][0].numel() - 1
        levels = (_lib.EncoderLevel * n_levels)()
        lim = None
        for l, e in enumerate(levels):
            pts, conv, pool = meta['points'][l], meta['_neighbors_i32'][l], (meta['_pools_i32'][l] if '_pools_i32' in meta else None)
            e.points, e.n = _lib.ptr(pts), int(pts.shape[0])
            has_conv = conv is not None and conv.shape[0] == pts.shape[0] and pts.shape[0] > 0
            e.conv_idx = _lib.iptr(conv) if has_conv else None
            has_pool = pool is not None and l + 1 < n_levels and pool.shape[0] == meta['points'][l + 1].shape[0] and pool.shape[0] > 0
            e.pool_idx = _lib.iptr(pool) if has_pool else None
            e.K = int(conv.shape[1]) if has_conv else (int(pool.shape[1]) if has_pool else 1)
            if has_conv and has_pool and conv.shape[1] != pool.shape[1]:
                return None
            e.pool_width = int(meta['_pool_width'][l]) if ('_pool_width' in meta and has_pool) else e.K
            e.seg_off, e.max_len = _lib.iptr(meta['_seg_off'][l]), int(max(meta['_lens_host'][l]))
        blocks = self._block_table()
        nb_blocks = len(self.encoder_blocks)
        if not L.regtr_encoder_supported(blocks, nb_blocks, levels, n_levels, n_clouds, start, stop):
            return None
        ctx = context.current()
        f16 = 1 if (ctx.f16_pair and not ctx.force_x3) else 0
        nb = L.regtr_encoder_ws_bytes(blocks, nb_blocks, levels, n_levels, n_clouds, start, stop, f16)
        if nb == 0:
            return None
        last = self.encoder_blocks[stop - 1]
        lq = last.layer_ind + (1 if 'strided' in last.block_name else 0)
        out_dim = last.KPConv.out_channels if isinstance(last, SimpleBlock) else last.unary2.out_dim
        out = torch.empty((meta['points'][lq].shape[0], out_dim), dtype=torch.float32, device=x.device)
        ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
        _lib.check(L.regtr_encoder_fwd(blocks, nb_blocks, levels, n_levels, n_clouds, start, stop, _lib.ptr(x), _lib.ptr(out), f16, 0.1, 1e-5,
                                       _lib.bptr(ws), nb, ctx.status_ptr(), _lib.stream()), 'regtr_encoder_fwd')
        return out

    def level0_blocks(self):
        """Number of leading blocks that work on level 0 only (not strided): they need nothing but level 0's conv table."""
        n = 0
        for b in self.encoder_blocks:
            if getattr(b, 'layer_ind', None) != 0 or 'strided' in b.block_name or 'pool' in b.block_name:
                break
            n += 1
        return n
