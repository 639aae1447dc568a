"""Batched, streamed inference harness: what test.py / demo.py run around `RegTR.forward`.

Replaces, for inference only, the reference's Trainer.test loop (/root/reference/src/trainer.py:177-211) and
GenericRegModel.test_step / _save_3DMatch_log (models/generic_reg_model.py:130-158, 260-281):
  * B pairs per forward instead of one (conf test_batch_size: 1), next batch loaded by a worker thread into pinned
    memory and copied to the GPU on a side stream while the current batch computes;
  * no compute_loss, no tensorboard, no per-pair host synchronisation: poses stay on the device until the end of the set;
  * pairs sharded over ranks (one process per GPU) with a single RCCL gather of the poses; rank 0 writes the logs;
  * `est.log` blocks / `pred_transforms.npy` in the reference's exact formats, so the reference's own evaluation scripts
    (benchmark_predator / RPMNet eval) read them unchanged.
"""
import json
import os
import pickle
import queue
import threading
import time

import numpy as np
import torch

from .distributed import gather_poses, shard_pairs


# ------------------------------------------------------------------------------------------------------ point cloud files
def read_ply_xyz(fname):
    """Minimal PLY reader (ascii / binary_little_endian, vertex x y z as float or double) -- stands in for
    open3d.io.read_point_cloud in demo.py:145-147."""
    with open(fname, 'rb') as f:
        if f.readline().strip() != b'ply':
            raise AssertionError(f'{fname}: not a PLY file')
        fmt, n_vert, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise AssertionError(f'{fname}: truncated PLY header')
            tok = line.split()
            if not tok:
                continue
            if tok[0] == b'format':
                fmt = tok[1].decode()
            elif tok[0] == b'element':
                in_vertex = tok[1] == b'vertex'
                if in_vertex:
                    n_vert = int(tok[2])
            elif tok[0] == b'property' and in_vertex:
                props.append((tok[1].decode(), tok[2].decode()))
            elif tok[0] == b'end_header':
                break
        names = [p[1] for p in props]
        if not all(a in names for a in 'xyz'):
            raise AssertionError(f'{fname}: vertex element has no x/y/z')
        if fmt == 'ascii':
            rows = np.loadtxt(f, max_rows=n_vert, ndmin=2)
            return rows[:, [names.index(a) for a in 'xyz']].astype(np.float64)
        if fmt != 'binary_little_endian':
            raise AssertionError(f'{fname}: unsupported PLY format {fmt}')
        np_t = {'float': '<f4', 'float32': '<f4', 'double': '<f8', 'float64': '<f8', 'uchar': 'u1', 'uint8': 'u1', 'char': 'i1',
                'int8': 'i1', 'short': '<i2', 'int16': '<i2', 'ushort': '<u2', 'uint16': '<u2', 'int': '<i4', 'int32': '<i4',
                'uint': '<u4', 'uint32': '<u4'}
        dt = np.dtype([(n, np_t[t]) for t, n in props])
        data = np.frombuffer(f.read(n_vert * dt.itemsize), dtype=dt, count=n_vert)
        return np.stack([data['x'], data['y'], data['z']], 1).astype(np.float64)


def load_point_cloud(fname, cache_dir=None):
    """demo.py:142-153: .pth (torch-saved array), .ply, .bin (KITTI float32 x y z r); returns (N, 3).
    cache_dir: `.pth` fragments (a zip + pickle: ~2 ms each to open, 3562 of them in the 3DLoMatch list) are mirrored there once as
    float32 `.npy` files and read back with np.load (~0.1 ms); a cache entry older than its source is rebuilt."""
    if cache_dir is not None and fname.endswith('.pth'):
        cname = os.path.join(cache_dir, os.path.abspath(fname).strip(os.sep).replace(os.sep, '__')[:-4] + '.npy')
        try:
            if os.path.getmtime(cname) >= os.path.getmtime(fname):
                return np.load(cname)
        except OSError:
            pass
        data = np.ascontiguousarray(load_point_cloud(fname), dtype=np.float32)
        os.makedirs(cache_dir, exist_ok=True)
        tmp = f'{cname}.{os.getpid()}.tmp.npy'
        np.save(tmp, data)
        os.replace(tmp, cname)                       # atomic: concurrent loader processes may race for the same entry
        return data
    if fname.endswith('.pth'):
        data = torch.load(fname, weights_only=False)
        data = data.numpy() if isinstance(data, torch.Tensor) else np.asarray(data)
    elif fname.endswith('.ply'):
        data = read_ply_xyz(fname)
    elif fname.endswith('.bin'):
        data = np.fromfile(fname, dtype=np.float32).reshape(-1, 4)
    else:
        raise AssertionError('Cannot recognize point cloud format')
    return data[:, :3]


# ------------------------------------------------------------------------------------------------------ pair sources
class ThreeDMatchPairs:
    """The 3DMatch / 3DLoMatch test pairs (data_loaders/threedmatch.py:20-101, test phase): an info pickle with keys
    rot (3,3), trans (3,1), src, tgt (paths relative to `root`), overlap."""

    def __init__(self, info_fname, root, cache_dir=None):
        with open(info_fname, 'rb') as fid:
            self.infos = pickle.load(fid)
        self.root, self.cache_dir = root, cache_dir

    def __len__(self):
        return len(self.infos['rot'])

    def __getitem__(self, i):
        src_path, tgt_path = self.infos['src'][i], self.infos['tgt'][i]
        pose = np.concatenate([self.infos['rot'][i], self.infos['trans'][i].reshape(3, 1)], 1).astype(np.float32)
        return {'src_xyz': np.ascontiguousarray(load_point_cloud(os.path.join(self.root, src_path), self.cache_dir), dtype=np.float32),
                'tgt_xyz': np.ascontiguousarray(load_point_cloud(os.path.join(self.root, tgt_path), self.cache_dir), dtype=np.float32),
                'pose': pose, 'idx': i, 'src_path': src_path, 'tgt_path': tgt_path}


class SyntheticPairs:
    """N deterministic synthetic 3DMatch-sized pairs (regtr_amd/synthetic.py) laid out like the 3DMatch test set
    (scene folders, cloud_bin_<k>.pth names) so that the est.log writer is exercised unchanged."""

    def __init__(self, n, points=20000, pairs_per_scene=64, overlap=None):
        self.n, self.points, self.pps, self.overlap = n, points, pairs_per_scene, overlap

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        from .synthetic import synth_pair
        src, tgt, pose = synth_pair(i, self.points, return_pose=True, overlap=self.overlap)
        sp, tp = _synthetic_paths(self, i)
        return {'src_xyz': src, 'tgt_xyz': tgt, 'pose': pose, 'idx': i, 'src_path': sp, 'tgt_path': tp}


def _materialize_one(args):
    root, n, points, overlap, distinct, i = args
    src = SyntheticPairs(n, points, overlap=overlap)
    it = src[i % distinct if distinct else i]
    for rel, arr in zip(_synthetic_paths(src, i), (it['src_xyz'], it['tgt_xyz'])):
        os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
        torch.save(arr, os.path.join(root, rel))
    return i, it['pose']


def materialize_synthetic(root, n, points=20000, overlap=None, logger=None, rank=0, world=1, distinct=0, procs=None):
    """Writes N synthetic pairs in the layout of the 3DMatch test set -- <root>/test/<scene>/cloud_bin_<k>.pth (torch-saved (N,3) float32
    arrays, what data_loaders/threedmatch.py:74-75 torch.load()s) and an info pickle (keys rot, trans, src, tgt, overlap,
    threedmatch.py:34-40) -- so that ThreeDMatchPairs, the loader thread and the est.log writer run exactly as on the real data set.
    Rank r generates and writes pairs r, r + world, ...; the ground-truth poses are exchanged through small per-rank files, so the
    pickle is complete on every rank.  distinct > 0: only that many different pairs are generated (pair i = pair i % distinct; every pair still
    gets its own two files, so the loader's work is the full set's) -- set-up time of the end-to-end measurement.  The pairs are generated
    by `procs` processes (default: the usable cores, at most 16; spawned, so the caller may already hold a GPU).  -> info pickle path"""
    src = SyntheticPairs(n, points, overlap=overlap)
    t0 = time.perf_counter()
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, f'test_info.rank{rank}.pkl')
    stamp = os.path.join(root, f'complete.rank{rank}.json')           # written last: (n, points, overlap, world) of a finished set
    want = json.dumps([n, points, overlap, world, distinct])
    if world == 1 and os.path.exists(stamp) and open(stamp).read() == want and os.path.exists(path):
        if logger:
            logger.info(f'{n} synthetic pairs already materialised under {root}')
        return path
    poses = {}
    todo = [(root, n, points, overlap, distinct, i) for i in range(rank, n, world)]
    procs = procs if procs is not None else max(1, min(16, len(os.sched_getaffinity(0)) // max(world, 1)))
    if procs > 1 and len(todo) > 4:
        import multiprocessing as mp
        with mp.get_context('spawn').Pool(procs) as pool:
            for i, pose in pool.imap_unordered(_materialize_one, todo, chunksize=max(1, len(todo) // (8 * procs))):
                poses[i] = pose
    else:
        for a in todo:
            i, pose = _materialize_one(a)
            poses[i] = pose
    with open(os.path.join(root, f'poses.rank{rank}.pkl'), 'wb') as f:
        pickle.dump(poses, f)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                      # every rank's files exist before anyone reads
    for r in range(world):
        if r != rank:
            with open(os.path.join(root, f'poses.rank{r}.pkl'), 'rb') as f:
                poses.update(pickle.load(f))
    infos = {'rot': [], 'trans': [], 'src': [], 'tgt': [], 'overlap': []}
    for i in range(n):
        sp, tp = _synthetic_paths(src, i)
        infos['rot'].append(poses[i][:, :3].astype(np.float64)); infos['trans'].append(poses[i][:, 3:].astype(np.float64))
        infos['src'].append(sp); infos['tgt'].append(tp); infos['overlap'].append(0.0)
    with open(path, 'wb') as f:
        pickle.dump(infos, f)
    with open(stamp, 'w') as f:
        f.write(want)
    if logger:
        logger.info(f'{n} synthetic pairs materialised under {root} in {time.perf_counter() - t0:.1f} s')
    return path


def _synthetic_paths(src, i):
    scene, k = i // src.pps, i % src.pps
    return (f'test/synthetic-scene{scene:03d}/cloud_bin_{2 * k + 1}.pth', f'test/synthetic-scene{scene:03d}/cloud_bin_{2 * k}.pth')


# ------------------------------------------------------------------------------------------------------ streaming
class Prefetcher:
    """Iterates `indices` of `pairs` in batches of `batch`; a worker thread loads the next batches (disk / generator,
    pinned host buffers), the H2D copies run on a side stream, and the consumer stream waits on an event only."""

    def __init__(self, pairs, indices, batch, device, depth=2):
        self.pairs, self.indices, self.batch, self.device = pairs, list(indices), batch, device
        self.q = queue.Queue(maxsize=depth)
        self.copy_stream = torch.cuda.Stream(device=device) if device.type == 'cuda' else None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def _work(self):
        try:
            for s in range(0, len(self.indices), self.batch):
                items = [self.pairs[i] for i in self.indices[s:s + self.batch]]
                out = {'items': items}
                if self.copy_stream is not None:
                    with torch.cuda.stream(self.copy_stream):
                        for key in ('src_xyz', 'tgt_xyz'):
                            out[key] = [torch.from_numpy(it[key]).pin_memory().to(self.device, non_blocking=True) for it in items]
                        out['ready'] = torch.cuda.Event()
                        out['ready'].record(self.copy_stream)
                else:
                    for key in ('src_xyz', 'tgt_xyz'):
                        out[key] = [torch.from_numpy(it[key]) for it in items]
                self.q.put(out)
            self.q.put(None)
        except BaseException as e:      # surface loader errors in the consumer
            self.q.put(e)

    def __iter__(self):
        while True:
            b = self.q.get()
            if b is None:
                return
            if isinstance(b, BaseException):
                raise b
            if 'ready' in b:
                torch.cuda.current_stream(self.device).wait_event(b['ready'])
                for key in ('src_xyz', 'tgt_xyz'):      # the tensors were allocated on the copy stream
                    for t in b[key]:
                        t.record_stream(torch.cuda.current_stream(self.device))
            yield b


# ------------------------------------------------------------------------------------------------------ process-pool loading
# The reference feeds its test loop from a torch DataLoader with `--num_workers` processes (test.py:26, data_loaders/__init__.py:11-58),
# one pair per step.  Here a forward takes 64 pairs in ~27 ms, i.e. the loader has to deliver ~2300 pairs/s = 4600 files/s, and a
# torch-saved fragment takes ~2 ms to open: one Python thread (Prefetcher) tops out at ~450 pairs/s.  BatchLoader therefore
#   * runs `workers` loader PROCESSES (fork: they inherit the pair source, never touch the GPU), each assembling WHOLE batches;
#   * gives every in-flight batch one slab of shared memory, page-locked once in the parent (hipHostRegister), into which the worker
#     writes the batch's clouds back to back in the forward's own order [src_0 .. src_{B-1}, tgt_0 .. tgt_{B-1}] -- so a batch crosses
#     PCIe as ONE asynchronous copy on a side stream instead of 128 small ones, and no array is pickled through a pipe;
#   * hands the consumer per-cloud VIEWS of that one device buffer; the consumer stream waits on the copy's event only.
_POOL_SOURCE = None          # the pair source of this process's loader workers (set before the fork)
_POOL_SLABS = None           # name -> numpy view of the shared slabs (inherited by the fork)


def _pool_fill(task):
    """Runs in a loader process: loads the pairs `idxs` and packs them into slab `slab_id`.  -> (b, slab_id, lens [2B], ids, None) or,
    when the batch does not fit the slab, (b, slab_id, None, ids, arrays) with the clouds pickled back (correct, slower)."""
    b, idxs, slab_id = task
    items = [_POOL_SOURCE[i] for i in idxs]
    clouds = [it['src_xyz'] for it in items] + [it['tgt_xyz'] for it in items]
    lens = [int(c.shape[0]) for c in clouds]
    ids = [int(it['idx']) for it in items]
    slab = _POOL_SLABS[slab_id]
    if sum(lens) > slab.shape[0]:
        return b, slab_id, None, ids, [np.ascontiguousarray(c, dtype=np.float32) for c in clouds]
    o = 0
    for c, n in zip(clouds, lens):
        slab[o:o + n] = c
        o += n
    return b, slab_id, lens, ids, None


class BatchLoader:
    """Iterates `indices` of `pairs` in batches of `batch` like Prefetcher, with `workers` loader processes and one pinned slab + one H2D
    copy per batch (see above).  Yields {'src_xyz': [...], 'tgt_xyz': [...], 'ids': [...]} with device tensors (CPU tensors for a cpu
    `device`: the tests' path).  slab_points: capacity of a slab in points (default: 1.5 x batch x 2 x 30k, grown never -- a larger batch
    comes back pickled)."""

    def __init__(self, pairs, indices, batch, device, workers=4, depth=None, slab_points=None):
        import multiprocessing as mp
        global _POOL_SOURCE, _POOL_SLABS
        self.indices, self.batch, self.device = list(indices), int(batch), device
        self.n_batches = (len(self.indices) + self.batch - 1) // self.batch
        self.workers = max(1, int(workers))
        depth = depth or (self.workers + 2)
        cap = int(slab_points or 1.5 * self.batch * 2 * 30000)
        self.cuda = device.type == 'cuda'
        # shared, page-locked slabs: anonymous shared mappings created BEFORE the fork (torch tensors in shared memory)
        self.slab_t = [torch.empty((cap, 3), dtype=torch.float32).share_memory_() for _ in range(min(depth, max(self.n_batches, 1)))]
        self.registered = []
        if self.cuda:
            rt = torch.cuda.cudart()
            for t in self.slab_t:
                if int(rt.cudaHostRegister(t.data_ptr(), t.numel() * 4, 0)) == 0:
                    self.registered.append(t.data_ptr())
        self.pinned = self.cuda and len(self.registered) == len(self.slab_t)
        self.stage = None if (self.pinned or not self.cuda) else torch.empty((cap, 3), dtype=torch.float32).pin_memory()
        _POOL_SOURCE, _POOL_SLABS = pairs, [t.numpy() for t in self.slab_t]
        self.pool = mp.get_context('fork').Pool(self.workers)
        _POOL_SOURCE = None
        self.copy_stream = torch.cuda.Stream(device=device) if self.cuda else None
        self.free = queue.Queue()
        for sid in range(len(self.slab_t)):
            self.free.put(sid)
        self.pending = queue.Queue()                  # AsyncResults in batch order
        self.out = queue.Queue(maxsize=len(self.slab_t))
        self.t_dispatch = threading.Thread(target=self._dispatch, daemon=True)
        self.t_upload = threading.Thread(target=self._upload, daemon=True)
        self.t_dispatch.start(); self.t_upload.start()

    def _dispatch(self):
        try:
            for b in range(self.n_batches):
                sid = self.free.get()
                idxs = self.indices[b * self.batch:(b + 1) * self.batch]
                self.pending.put(self.pool.apply_async(_pool_fill, ((b, idxs, sid),)))
            self.pending.put(None)
        except BaseException as e:      # noqa: BLE001
            self.pending.put(e)

    def _upload(self):
        try:
            while True:
                res = self.pending.get()
                if res is None:
                    break
                if isinstance(res, BaseException):
                    raise res
                b, sid, lens, ids, arrays = res.get()
                if arrays is not None:                      # oversize batch: the clouds came back by value
                    lens = [int(a.shape[0]) for a in arrays]
                    host = torch.from_numpy(np.concatenate(arrays))
                    if self.cuda:
                        host = host.pin_memory()
                else:
                    host = self.slab_t[sid][:sum(lens)]
                B = len(ids)
                off = np.concatenate([[0], np.cumsum(lens)])
                out = {'ids': ids}
                if self.cuda:
                    with torch.cuda.stream(self.copy_stream):
                        if arrays is None and not self.pinned:      # (hipHostRegister unavailable: through one pinned staging buffer)
                            self.stage[:host.shape[0]].copy_(host)
                            host = self.stage[:host.shape[0]]
                        dev = host.to(self.device, non_blocking=True)
                        ready = torch.cuda.Event()
                        ready.record(self.copy_stream)
                    ready.synchronize()                     # the slab (and the staging buffer) may be refilled from here on
                    out['dev'] = dev
                else:
                    dev = host.clone()
                self.free.put(sid)
                views = [dev[off[c]:off[c + 1]] for c in range(2 * B)]
                out['src_xyz'], out['tgt_xyz'] = views[:B], views[B:]
                self.out.put(out)
            self.out.put(None)
        except BaseException as e:      # noqa: BLE001  (surface loader errors in the consumer)
            self.out.put(e)

    def __iter__(self):
        try:
            while True:
                b = self.out.get()
                if b is None:
                    return
                if isinstance(b, BaseException):
                    raise b
                if self.cuda:
                    b['dev'].record_stream(torch.cuda.current_stream(self.device))      # allocated on the copy stream, consumed here
                yield b
        finally:
            self.close()

    def close(self):
        if self.pool is not None:
            self.pool.terminate(); self.pool.join()
            self.pool = None
            if self.registered:
                rt = torch.cuda.cudart()
                for p in self.registered:
                    rt.cudaHostUnregister(p)
                self.registered = []


# ------------------------------------------------------------------------------------------------------ result files
_POSE_FMT = ('\t'.join(['%.12f'] * 4) + '\n') * 4      # 4 rows, tab-separated, 12 decimals (generic_reg_model.py:279-281)


def write_est_log(log_path, benchmark, records, append=False):
    """generic_reg_model.py:260-281: per scene `<log_path>/<benchmark>/<scene>/est.log`, one block per pair:
    "{tgt_idx}\\t{src_idx}\\t-1" then the 4x4 pose, rows tab-separated with 12 decimals.
    A run writes each scene's file once, so the file is TRUNCATED unless append=True -- the reference opens it in append
    mode per pair (:276), which duplicates blocks (and corrupts the recall) when a run is repeated into the same log folder."""
    by_scene = {}
    for rec in records:
        scene = rec['src_path'].split(os.path.sep)[1]
        by_scene.setdefault(scene, []).append(rec)
    for scene, recs in by_scene.items():
        folder = os.path.join(log_path, benchmark, scene)
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, 'est.log'), 'a' if append else 'w') as fid:
            for rec in recs:
                src_idx = int(os.path.basename(rec['src_path']).split('_')[-1].replace('.pth', ''))
                tgt_idx = int(os.path.basename(rec['tgt_path']).split('_')[-1].replace('.pth', ''))
                pose = np.asarray(rec['pose'], dtype=np.float64)
                if pose.shape[0] == 3:
                    pose = np.concatenate([pose, [[0., 0., 0., 1.]]], axis=0)
                fid.write('{}\t{}\t{}\n'.format(tgt_idx, src_idx, -1) + _POSE_FMT % tuple(pose.ravel()))


def pose_errors(pred, gt):
    """Rotation error (deg) and translation error of (n, 3, 4) predictions against ground truth
    (utils/se3_torch.py se3_compare / generic_reg_model.py:198-210)."""
    pred, gt = np.asarray(pred, np.float64), np.asarray(gt, np.float64)
    Rd = np.einsum('nij,nkj->nik', pred[:, :, :3], gt[:, :, :3])
    tr = np.clip((np.trace(Rd, axis1=1, axis2=2) - 1) / 2, -1, 1)
    return np.degrees(np.arccos(tr)), np.linalg.norm(pred[:, :, 3] - gt[:, :, 3], axis=1)


# ------------------------------------------------------------------------------------------------------ the test loop
def run_test(model, pairs, batch, device, logger=None, max_pairs=None, num_workers=0):
    """Runs every pair of `pairs` (this rank's shard) through the model, B at a time.  num_workers > 0: loader processes + one pinned slab
    and one H2D copy per batch (BatchLoader); 0: one loader thread (Prefetcher).  Returns, on every rank,
    (poses (n_total, 3, 4) float32 numpy ordered by pair id, pair ids, timing dict)."""
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    n = len(pairs) if max_pairs is None else min(len(pairs), max_pairs)
    mine = shard_pairs(n, rank, world)
    model.eval()
    poses, ids = [], []
    t0 = time.perf_counter()
    loader = BatchLoader(pairs, mine, batch, device, workers=num_workers) if num_workers > 0 else Prefetcher(pairs, mine, batch, device)
    with torch.no_grad():
        for b in loader:
            out = model({'src_xyz': b['src_xyz'], 'tgt_xyz': b['tgt_xyz']})
            poses.append(out['pose'][-1])                         # (B, 3, 4), stays on the device
            ids.extend(b['ids'] if 'ids' in b else [it['idx'] for it in b['items']])
    pose_t = torch.cat(poses).reshape(-1, 12) if poses else torch.zeros((0, 12), dtype=torch.float32, device=device)
    id_t = torch.tensor(ids, dtype=torch.int32, device=device)
    all_poses, all_ids = gather_poses(pose_t, id_t, n)
    if device.type == 'cuda':
        torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if logger and rank == 0:
        logger.info(f'{n} pairs on {world} GPU(s) in {elapsed:.2f} s = {n / elapsed:.1f} pairs/s (incl. loading)')
    poses_np = all_poses.reshape(-1, 3, 4).cpu().numpy()
    if not np.isfinite(poses_np).all():
        bad = np.unique(np.nonzero(~np.isfinite(poses_np))[0])
        raise RuntimeError(f'{len(bad)} of {len(poses_np)} predicted poses are not finite (first pair ids: {all_ids.cpu().numpy()[bad[:5]].tolist()}).  '
                           "(An f16 pair operand beyond 65504 is not the cause unless cfg.f16_range_check was switched off: RegTR.forward detects that "
                           "and re-runs the forward in fp32x3 arithmetic; compute_dtype: 'fp32x3' avoids the format altogether.)  Check the inputs "
                           'and the checkpoint for non-finite values.')
    return poses_np, all_ids.cpu().numpy(), {'elapsed_s': elapsed, 'pairs': n, 'world': world}
