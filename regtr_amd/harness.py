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


def usable_cores():
    """Host cores this process may actually use: the affinity mask capped by a cgroup CPU quota (os.cpu_count() reports the machine's
    cores even inside a quota-limited container; thread teams larger than the quota spin against each other until the kernel throttles
    the whole process)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            n = min(n, max(1, q // per))
    except (OSError, ValueError):
        pass
    return n


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
    if procs > 1 and len(todo) >= 32:        # (spawned workers import torch: seconds each -- not worth it for a handful of pairs)
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
        self.copy_stream = torch.cuda.Stream(device=device, priority=-1) if device.type == 'cuda' else None      # (own hardware-queue pool, see LoaderPool.iterate)
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
# torch-saved fragment takes ~2 ms to open: one Python thread (Prefetcher) tops out at ~450-530 pairs/s.  LoaderPool therefore
#   * runs `workers` loader PROCESSES (forked once, reusable over any number of passes; they inherit the pair source and never touch
#     the GPU), each assembling WHOLE batches;
#   * gives every in-flight batch one slab of shared memory (anonymous shared mappings: no page is touched until a worker fills it)
#     into which the worker writes the batch's clouds back to back in the forward's own order [src_0 .. src_{B-1}, tgt_0 .. tgt_{B-1}]
#     -- no array is pickled through a pipe;
#   * moves a filled slab through one of two page-locked staging buffers to the GPU as ONE asynchronous copy on a side stream instead of
#     128 small ones, and hands the consumer per-cloud VIEWS of that one device buffer; the consumer stream waits on the copy's event.
# Measured on the 16-core-quota GPU boxes (profiles/r04_*_e2e_harness.txt): more than ~6 loader processes slow the set down (they compete
# with the launching thread for the quota), hence the default of 4.
_POOL_SLABS = {}             # pool id -> [numpy views of the pool's shared slabs]; set before the fork and kept until close(), so that a
#                              worker multiprocessing.Pool re-spawns later (forked from the parent THEN) finds them too
_POOL_SOURCES = {}           # (worker side) source file -> unpickled pair source, loaded once per worker
_PINNED = {}                 # (device, points) -> two page-locked staging buffers, kept for the life of the process
_POOL_IDS = [0]


def _pool_fill(task):
    """Runs in a loader process: loads the pairs `idxs` (one PART of a batch) and packs them into its region [lo, lo + cap) of slab
    `slab_id` as [src clouds ..., tgt clouds ...].  -> (b, part, lens_src, lens_tgt, ids, None) or, when the part does not fit its
    region, (b, part, None, None, ids, (src arrays, tgt arrays)) with the clouds pickled back (correct, slower)."""
    b, part, idxs, slab_id, lo, cap, pool_id, source_file = task
    source = _POOL_SOURCES.get(source_file)
    if source is None:        # first task of this worker (or of a re-spawned one): the pair source arrives by file, not by fork
        import pickle
        with open(source_file, 'rb') as f:
            source = pickle.load(f)
        _POOL_SOURCES.clear()
        _POOL_SOURCES[source_file] = source
    items = [source[i] for i in idxs]
    src, tgt = [it['src_xyz'] for it in items], [it['tgt_xyz'] for it in items]
    ls, lt = [int(c.shape[0]) for c in src], [int(c.shape[0]) for c in tgt]
    ids = [int(it['idx']) for it in items]
    if sum(ls) + sum(lt) > cap:
        return b, part, None, None, ids, ([np.ascontiguousarray(c, dtype=np.float32) for c in src], [np.ascontiguousarray(c, dtype=np.float32) for c in tgt])
    slab = _POOL_SLABS[pool_id][slab_id]
    o = lo
    for c, n in zip(src + tgt, ls + lt):
        slab[o:o + n] = c
        o += n
    return b, part, ls, lt, ids, None


def _unlink_quiet(path):
    try:
        os.unlink(path)
    except OSError:
        pass


class LoaderPool:
    """`workers` forked loader processes + their shared slabs over one pair source; `iterate(indices, batch)` yields batches
    {'src_xyz': [...], 'tgt_xyz': [...], 'ids': [...]} of device tensors (CPU tensors for a cpu `device`: the tests' path), in order.
    The pool outlives a pass: create it once, iterate as often as needed, close() at the end.  It can (and test.py does) be created
    BEFORE the GPU is initialised and before torch.distributed -- forking a process that holds a HIP context / a process group copies that
    state into every worker: the constructor touches no GPU API, the pair source may be handed over later (`set_source`, by a pickle file:
    workers -- including ones multiprocessing re-spawns -- load it on their first task) and the page-locked staging buffers are made by
    `prepare()` / the first pass.  slab_points: capacity of a slab in points (default 1.25 x max_batch x 2 x 24k; a batch that does not
    fit comes back pickled -- correct, slower)."""

    def __init__(self, pairs, device, workers=4, max_batch=64, depth=None, slab_points=None, maxtasksperchild=None):
        import mmap
        import multiprocessing as mp
        self.device = torch.device(device)
        self.workers = max(1, int(workers))
        self.cuda = self.device.type == 'cuda'
        self.cap = int(slab_points or 1.25 * max_batch * 2 * 24000)
        n_slabs = depth or (self.workers + 2)
        self._maps = [mmap.mmap(-1, self.cap * 12) for _ in range(n_slabs)]           # MAP_SHARED | MAP_ANONYMOUS: inherited by the fork
        self.slabs = [np.frombuffer(m, dtype=np.float32).reshape(self.cap, 3) for m in self._maps]
        _POOL_IDS[0] += 1
        self.pool_id = _POOL_IDS[0]
        _POOL_SLABS[self.pool_id] = self.slabs          # stays registered until close(): re-spawned workers inherit it as well
        self.pool = mp.get_context('fork').Pool(self.workers, maxtasksperchild=maxtasksperchild)
        self._source_file = None
        self.copy_stream = None
        self.last_timing = None
        if pairs is not None:
            self.set_source(pairs)
        if self.cuda and torch.cuda.is_initialized():
            self.prepare()

    def set_source(self, pairs):
        """The pair source the workers index (`pairs[i]` -> {'src_xyz', 'tgt_xyz', 'idx', ...}); must pickle."""
        import pickle
        import tempfile
        import weakref
        self._drop_source_file()
        fd, self._source_file = tempfile.mkstemp(prefix='regtr_pairs_', suffix='.pkl')
        with os.fdopen(fd, 'wb') as f:
            pickle.dump(pairs, f, protocol=pickle.HIGHEST_PROTOCOL)
        # sys.exit() / an exception / an abandoned pool: the file is removed when the pool is collected or the interpreter exits
        self._source_finalizer = weakref.finalize(self, _unlink_quiet, self._source_file)

    def _drop_source_file(self):
        if self._source_file is not None:
            fin = getattr(self, '_source_finalizer', None)
            if fin is not None:
                fin.detach()
            _unlink_quiet(self._source_file)
            self._source_file = None

    def prepare(self):
        """Page-locks the two staging buffers (tens of ms: not inside the first timed pass).  Needs the GPU runtime, so call it after
        torch.cuda.set_device when the pool was created before."""
        if self.cuda:
            self._staging()

    def _staging(self):
        key = (str(self.device), self.cap)
        if key not in _PINNED:
            _PINNED[key] = [torch.empty((self.cap, 3), dtype=torch.float32).pin_memory() for _ in range(2)]
        return _PINNED[key]

    def iterate(self, indices, batch):
        indices, batch = list(indices), int(batch)
        n_batches = (len(indices) + batch - 1) // batch
        free, pending, out = queue.Queue(), queue.Queue(), queue.Queue(maxsize=len(self.slabs))
        for sid in range(len(self.slabs)):
            free.put(sid)
        if self.cuda and self.copy_stream is None:
            # high priority = a hardware queue from another pool than the forward's stream (regtr.py: _side_stream; an RCCL communicator in
            # the process had put a normal-priority side stream onto the main stream's queue, i.e. behind the forward it should run next to)
            self.copy_stream = torch.cuda.Stream(device=self.device, priority=-1)
        if self._source_file is None:
            raise RuntimeError('LoaderPool.iterate: no pair source (set_source)')
        import sys
        interval = sys.getswitchinterval()
        if interval > 0.0005:
            sys.setswitchinterval(0.0005)    # the launching thread shares the interpreter with the two loader threads below: short GIL
        #                                      hand-offs for the duration of the pass (restored when the generator finishes)
        stage = self._staging() if self.cuda else None
        timing = {'wait_worker_s': 0.0, 'stage_copy_s': 0.0, 'h2d_wait_s': 0.0, 'batches': n_batches}
        self.last_timing = timing

        # a batch is loaded by up to `workers` processes at once: part p fills region p of the batch's slab (regions of equal capacity;
        # the copy into the staging buffer compacts them into the forward's order) -- the latency of the FIRST batch is what a short
        # set pays in full
        parts = max(1, min(self.workers, 8, batch // 8))
        region = self.cap // parts

        def dispatch():
            try:
                for b in range(n_batches):
                    sid = free.get()
                    idxs = indices[b * batch:(b + 1) * batch]
                    per = (len(idxs) + parts - 1) // parts
                    tasks = [(b, p, idxs[p * per:(p + 1) * per], sid, p * region, region, self.pool_id, self._source_file)
                             for p in range(parts) if idxs[p * per:(p + 1) * per]]
                    pending.put((sid, [self.pool.apply_async(_pool_fill, (t,)) for t in tasks]))
                pending.put(None)
            except BaseException as e:      # noqa: BLE001
                pending.put(e)

        def upload():
            try:
                staged = [None, None]                       # event behind the last H2D copy out of each staging buffer
                k = 0
                while True:
                    res = pending.get()
                    if res is None:
                        break
                    if isinstance(res, BaseException):
                        raise res
                    t0 = time.perf_counter()
                    sid, asyncs = res
                    done = [a.get() for a in asyncs]                # parts in order
                    t1 = time.perf_counter()
                    timing['wait_worker_s'] += t1 - t0
                    # pieces in the forward's order: every part's src clouds, then every part's tgt clouds
                    pieces, lens_s, lens_t, ids = [[], []], [], [], []
                    for (b, part, ls, lt, pid, arrays), t in zip(done, [a_ for a_ in range(len(done))]):
                        ids += pid
                        if arrays is not None:              # oversize part: the clouds came back by value
                            ls, lt = [int(a.shape[0]) for a in arrays[0]], [int(a.shape[0]) for a in arrays[1]]
                            pieces[0] += [torch.from_numpy(a) for a in arrays[0]]
                            pieces[1] += [torch.from_numpy(a) for a in arrays[1]]
                        else:
                            lo = part * region
                            pieces[0].append(torch.from_numpy(self.slabs[sid][lo:lo + sum(ls)]))
                            pieces[1].append(torch.from_numpy(self.slabs[sid][lo + sum(ls):lo + sum(ls) + sum(lt)]))
                        lens_s += ls; lens_t += lt
                    lens = lens_s + lens_t
                    n = sum(lens)
                    B = len(ids)
                    off = np.concatenate([[0], np.cumsum(lens)])
                    item = {'ids': ids}
                    if self.cuda:
                        if n <= self.cap:
                            if staged[k] is not None:
                                staged[k].synchronize()     # this staging buffer's previous copy has left
                            pin = stage[k][:n]
                        else:
                            pin = torch.empty((n, 3), dtype=torch.float32).pin_memory()
                        # plain memcpy (numpy): a torch copy_ of this size wakes the whole intra-op thread team -- on a many-core host
                        # under a cgroup CPU quota the spinning team exhausts the quota and the kernel throttles EVERY thread of the
                        # process for the rest of the 100 ms period (40-70 ms stalls of the launching thread were measured that way)
                        pin_np = pin.numpy()
                        o = 0
                        for pc in pieces[0] + pieces[1]:
                            np.copyto(pin_np[o:o + pc.shape[0]], pc.numpy())
                            o += pc.shape[0]
                        t2 = time.perf_counter()
                        timing['stage_copy_s'] += t2 - t1
                        with torch.cuda.stream(self.copy_stream):
                            dev = pin.to(self.device, non_blocking=True)
                            ready = torch.cuda.Event()
                            ready.record(self.copy_stream)
                        if n <= self.cap:
                            staged[k] = ready
                            k ^= 1
                        else:
                            ready.synchronize()
                        item['dev'], item['ready'] = dev, ready
                    else:
                        dev = torch.cat(pieces[0] + pieces[1])
                    free.put(sid)                           # the slab is free as soon as its contents sit in the staging buffer
                    views = [dev[off[c]:off[c + 1]] for c in range(2 * B)]
                    item['src_xyz'], item['tgt_xyz'] = views[:B], views[B:]
                    out.put(item)
                out.put(None)
            except BaseException as e:      # noqa: BLE001  (surface loader errors in the consumer)
                out.put(e)

        threading.Thread(target=dispatch, daemon=True).start()
        threading.Thread(target=upload, daemon=True).start()
        try:
            while True:
                t0 = time.perf_counter()
                b = out.get()
                timing['h2d_wait_s'] += time.perf_counter() - t0      # (time the CONSUMER stood waiting for a batch)
                if b is None:
                    return
                if isinstance(b, BaseException):
                    raise b
                if self.cuda:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(b['ready'])
                    b['dev'].record_stream(cur)                 # allocated on the copy stream, consumed here
                yield b
        finally:
            sys.setswitchinterval(interval)                     # the process-wide switch interval is the caller's again

    def close(self):
        if self.pool is not None:
            self.pool.terminate(); self.pool.join()
            self.pool = None
        _POOL_SLABS.pop(self.pool_id, None)
        self._drop_source_file()
        self.slabs = []
        for m in self._maps:
            try:
                m.close()
            except BufferError:         # a numpy view still alive somewhere: the mapping goes with the process
                pass
        self._maps = []


class BatchLoader:
    """One pass of a LoaderPool created for it (and closed at the end): `for b in BatchLoader(pairs, indices, batch, device, workers)`."""

    def __init__(self, pairs, indices, batch, device, workers=4, depth=None, slab_points=None):
        self.pool = LoaderPool(pairs, device, workers=workers, max_batch=batch, depth=depth, slab_points=slab_points)
        self.indices, self.batch = indices, batch

    def __iter__(self):
        try:
            yield from self.pool.iterate(self.indices, self.batch)
        finally:
            self.pool.close()


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
def run_test(model, pairs, batch, device, logger=None, max_pairs=None, num_workers=0, loader_pool=None):
    """Runs every pair of `pairs` (this rank's shard) through the model, B at a time.  loader_pool: a LoaderPool over `pairs` (loader
    processes forked by the caller, e.g. before the GPU was initialised); else num_workers > 0: a pool forked here for this pass;
    0: one loader thread (Prefetcher).
    model: one RegTR module, or a LIST of replicas with the same weights (regtr_amd.workload.replicate): R host threads then take batches off the
    one loader in turn and run them on R HIP streams -- forwards in flight fill each other's host waits and heads (round 6; bench.py measures
    +6 % for three 64-pair forwards in flight); the poses come back in pair-id order either way, bit-identical to the one-replica run.
    Returns, on every rank, (poses (n_total, 3, 4) float32 numpy ordered by pair id, pair ids, timing dict)."""
    import contextlib
    import itertools
    import threading
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    n = len(pairs) if max_pairs is None else min(len(pairs), max_pairs)
    mine = shard_pairs(n, rank, world)
    models = list(model) if isinstance(model, (list, tuple)) else [model]
    for m in models:
        m.eval()
    t0 = time.perf_counter()
    if loader_pool is not None:
        loader = loader_pool.iterate(mine, batch)
    else:
        loader = BatchLoader(pairs, mine, batch, device, workers=num_workers) if num_workers > 0 else Prefetcher(pairs, mine, batch, device)
    cuda = device.type == 'cuda'
    R = len(models)
    streams = [torch.cuda.Stream(device) for _ in range(R)] if (cuda and R > 1) else [None] * R
    it, turn, counter = iter(loader), threading.Lock(), itertools.count()
    done, errs, fwd_ms = [], [], []

    def work(r):
        try:
            ctx = (torch.cuda.stream(streams[r]) if streams[r] is not None else contextlib.nullcontext())
            dctx = torch.cuda.device(device) if cuda else contextlib.nullcontext()
            with dctx, ctx, torch.no_grad():
                while not errs:
                    with turn:                                     # one loader, R consumers: batches are handed out in order
                        try:
                            b = next(it)
                        except StopIteration:
                            return
                        k = next(counter)
                    t_f = time.perf_counter()
                    out = models[r]({'src_xyz': b['src_xyz'], 'tgt_xyz': b['tgt_xyz']})
                    fwd_ms.append(round((time.perf_counter() - t_f) * 1e3, 1))
                    done.append((k, out['pose'][-1], b['ids'] if 'ids' in b else [i_['idx'] for i_ in b['items']]))      # (B, 3, 4), stays on the device
        except BaseException as e:      # noqa: BLE001  (re-raised on the caller)
            errs.append(e)
    if R == 1:
        work(0)
    else:
        cur = torch.cuda.current_stream(device) if cuda else None
        th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for s_ in streams:
            if s_ is not None:
                cur.wait_stream(s_)
        for _, p_, _ in done:
            if cuda:
                p_.record_stream(cur)
    if errs:
        raise errs[0]
    done.sort(key=lambda d: d[0])
    poses = [d[1] for d in done]
    ids = [i for d in done for i in d[2]]
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
    return poses_np, all_ids.cpu().numpy(), {'elapsed_s': elapsed, 'pairs': n, 'world': world,
                                             'loader': getattr(loader_pool, 'last_timing', None), 'forward_ms': fwd_ms}
