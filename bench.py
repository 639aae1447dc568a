"""Throughput benchmark of the RegTR correspondence-inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    N > 1: either launched by the driver as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py
    --gpus N ...`, or plainly as `python bench.py --gpus N` -- the script then re-executes itself under
    torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous).  A world size that differs from --gpus is an error,
    and `n_gpus` in the JSON line is the number of ranks whose poses arrived through the RCCL all_gather.

A "step" is one forward of the hot path (preprocess -> KPConv encoder -> 6 cross-attention layers -> head ->
weighted Procrustes) over one batch of `--pairs` synthetic 3DMatch-sized pairs (BASELINE.json configs[2]: ~20k points
per cloud, conf/3dmatch.yaml architecture, random-init weights), inputs resident in HBM.  Independent pairs shard across
ranks with no data-path collective; the only RCCL traffic is one all_gather of the poses at the end of the timed region.
Rank 0 prints ONE JSON line (metric pairs/s, whole-job aggregate) with `roofline` (KPConv gather, HBM bound) and
`cpu_baseline` (the CPU oracle port timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')   # CPU-baseline leg only: idle OpenMP workers must not burn a CPU quota

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_BF16_PEAK_TFS = 2500.0   # dense bf16 MFMA peak (same guide); the f32-input MFMA peak is 157.3


from regtr_amd.synthetic import synth_modelnet_pair, synth_pair  # noqa: E402  
from regtr_amd.workload import (DEFAULT_PAIRS, DEFAULT_REPLICAS, REAL_PAIRS, REDUCED_TOL, build_workload, kpconv_algorithmic_bytes, parity_slots,  # noqa: E402,F401
                                probe_head, real_pairs, ReplicaRunner, replicate)
from regtr_amd.measure import (code_version, collect_pmc, forward_traffic, measure_attention, measure_gemm_roofline, measure_kpconv_roofline,  # noqa: E402,F401
                               measure_preprocess, one_stream, pmc_traffic)


# ----------------------------------------------------------------------------------------------------------------
def measure_real_fragments(args, dev, dtype, steps=8, parity_pairs=4):
    """`real_fragments_pairs_per_s` of the default line: BASELINE configs[2] on the three REAL 3DMatch pairs the reference ships (demo.py:26-49; red-kitchen,
    hotel_umd, home_at), replicated to the same pairs per forward under random rigid motions (workload.real_pairs), probe head -- 2 warm-up + `steps` timed
    forwards between device synchronisations, then the CPU-oracle gate on `parity_pairs` of them."""
    R = max(1, args.replicas)
    n = R * args.pairs
    cfg_r, model_r, pairs_r, batch_r = build_workload('3dmatch', n, args.points, False, 0, dev, dtype, real=True)
    chunks_r = [(i * args.pairs, (i + 1) * args.pairs) for i in range(R)]
    runner = ReplicaRunner(replicate(model_r, cfg_r, R, dev), batch_r, chunks_r, dev)
    runner.run(2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = runner.run(steps)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    out, pairs_r = outs[0], pairs_r[:args.pairs]
    first = {k: v[:args.pairs] for k, v in batch_r.items()}
    lv = [int(p.shape[0]) for p in model_r.preprocessor(list(first['src_xyz']) + list(first['tgt_xyz']))['points']]
    info = {'value': n / dt, 'unit': 'pairs/s', 'ms_per_step': dt * 1e3, 'steps': steps, 'pairs_per_step': n, 'concurrent_forwards': R, 'level_points_per_forward': lv,
            'points_per_cloud': [int(np.mean([len(s) for s, _ in pairs_r])), int(np.mean([len(t) for _, t in pairs_r]))],
            'data': 'the three 3DMatch pairs the reference ships (tests/golden/3dmatch_*.npz), replicated under random rigid motions; head: linear probe'}
    if parity_pairs > 0:
        slots = parity_slots([len(a) + len(b) for a, b in pairs_r], parity_pairs)
        p = parity_check(cfg_r, model_r, pairs_r, out, slots)
        info['parity'] = {k: p[k] for k in ('ok', 'pose_max_abs', 'corr_max_abs', 'kabsch_cond_max', 'pairs_checked', 'keypoints_bit_exact', 'reason')}
    del model_r, batch_r, runner, outs
    torch.cuda.empty_cache()
    return info['value'], info


def usable_cores():
    """Host cores this process may actually use: affinity mask, capped by a cgroup CPU quota when there is one
    (os.cpu_count() reports the machine's cores even inside a quota-limited container, and OpenMP teams larger than
    the quota spin against each other)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()
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


def pick_cpu_threads(run_probe):
    """Thread count for the CPU baseline, chosen by MEASUREMENT: `run_probe()` (the real pipeline on a 1/4 crop) is timed
    at all, 1/2, 1/4 of the usable cores and 1 thread, and the fastest wins; the descent stops as soon as fewer threads
    are clearly slower.  (Under a cgroup CPU quota on a busy many-core host, OpenMP teams as large as the quota can be
    far slower than smaller ones, so "all cores" is not assumed.)"""
    cores = usable_cores()
    cands = sorted({max(1, cores), max(1, cores // 2), max(1, cores // 4), 1}, reverse=True)
    best, best_t = cands[0], float('inf')
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        run_probe()
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = c, dt
        elif dt > 1.3 * best_t:
            break
    torch.set_num_threads(best)
    return best, cores


def cpu_baseline(cfg, pairs, max_seconds=20.0, cfg_name='3dmatch'):
    """The reference's CPU path on this box's host cores, same workload, bounded sample: whole pairs are timed until
    `max_seconds` of CPU work have been spent and at least five pairs ran, after warm-up / thread selection on crops of the
    first pair.  kind 'reference' = the REAL reference RegTR module (imported from /root/reference, its CPU Preprocessor over the
    unmodified reference C++) where that tree exists; kind 'port' = the CPU oracle restatement pinned to it
    (oracle/regtr_ref.py; preprocessing through oracle/_ref when present) -- the GPU boxes have no /root/reference."""
    from oracle import native, ref_loader, regtr_ref, seeded_weights
    from regtr_amd.kernel_points import K015_CENTER
    sd = seeded_weights.seeded_state_dict(cfg, 0, K015_CENTER)
    use_ref = native.have_ref()
    kind = 'port'
    if ref_loader.available() and use_ref:
        rcfg = ref_loader.load_cfg(cfg_name)
        ref_model = ref_loader.build_model(rcfg, 0)
        ref_model.load_state_dict(sd, strict=True)
        kind = 'reference'

        def forward(s, t, timings=None):
            t0 = time.perf_counter()
            ref_model({'src_xyz': [torch.from_numpy(s)], 'tgt_xyz': [torch.from_numpy(t)]})
            if timings is not None:
                timings.append((float('nan'), float('nan'), time.perf_counter() - t0))
    else:
        def forward(s, t, timings=None):
            regtr_ref.regtr_forward(sd, cfg, [s], [t], use_ref_cpp=use_ref, timings=timings)
    times, stages = [], []
    with torch.no_grad():
        s0, t0_ = pairs[0]
        # rows are spatially ordered, so a prefix is a compact crop
        crop = lambda f: forward(s0[:max(len(s0) // f, 64)], t0_[:max(len(t0_) // f, 64)])
        crop(16)                                                                      # warm-up
        threads, cores = pick_cpu_threads(lambda: crop(4))
        t_start = time.perf_counter()
        for s, t in pairs:
            tm = []
            t0 = time.perf_counter()
            forward(s, t, tm)
            times.append(time.perf_counter() - t0); stages.append(tm[0])
            if time.perf_counter() - t_start > max_seconds and len(times) >= 5:
                break
    med = float(np.median(times))
    st = np.median(np.array(stages), axis=0)
    what = ('the REAL reference RegTR module (src/models/regtr.py) with its CPU Preprocessor over the unmodified reference C++' if kind == 'reference'
            else 'fp32 torch CPU restatement of the reference modules; preprocessing by ' +
                 ('the unmodified reference C++ (oracle/_ref)' if use_ref else 'the C++ oracle restatement'))
    split = '' if kind == 'reference' else f' = preprocess {st[0]:.2f} + encoder {st[1]:.2f} + attention/head/pose {st[2]:.2f}'
    return {'value': 1.0 / med, 'unit': 'pairs/s', 'cores': threads, 'kind': kind,
            'sample': f'{len(times)} timed pair(s) (bounded to ~{max_seconds:.0f} s of CPU work, >= 5 pairs) after warm-up, same synthetic '
                      f'~{len(pairs[0][0])}-pt pairs, {what}, on {threads} threads '
                      f'(fastest of 1, 1/4, 1/2, all of the {cores} usable cores on a 1/4-crop probe; os.cpu_count()={os.cpu_count()}); '
                      f'median s/pair {med:.3f}{split}'}


PARITY_TOL = 1e-4      # BASELINE.json north_star: "predicted correspondences and R|t within 1e-4 abs"


def parity_check(cfg, model, pairs, out, which, parity_mode=False):
    """BASELINE.json's metric is "pairs/sec ...; pose err vs ref": the outputs of the LAST TIMED STEP (`out`, the product's dict for
    the whole batch) for the pairs `which`, against the CPU oracle run on each of those pairs alone with the same weights --
    oracle/regtr_ref.py (pinned to the real reference module, tests/test_oracle.py) over the canonical tables built from the
    unmodified reference C++'s neighbour sets (oracle/canonical.py); in --parity-mode over the reference C++'s own orders.
    Quantities: /root/reference/src/models/regtr.py:185-235 (pose, correspondences, overlap logits), utils/se3_torch.py:108-154.
    The oracle is the CHECKER here (outside the timed region), never the thing measured.

    Gate: key points bit-exact, correspondences within 1e-4 AND R|t within 1e-4 of the oracle's, on every checked pair -- nothing
    else makes `ok` true.  With RANDOM-INIT weights (there are no checkpoints here) the predicted correspondences nearly collapse
    (spread 1.5 cm against 50 cm of key-point spread), so the Kabsch covariance is close to rank one -- singular values 0.5 / 1e-3 /
    2e-5 measured -- and R amplifies a 1e-6 correspondence difference by 1 / (s2 + s3) ~ 1e2; the reference's own float32 Kabsch then
    differs from a float64 solve of the SAME inputs by up to 1.6e-5 (`oracle_f32_vs_f64_kabsch`).  A pose above 1e-4 on such a pair is
    therefore DIAGNOSED, not excused: `ok` is false and `reason` says 'conditioning' when the correspondences are within 1e-4, the problem
    is ill-conditioned (s1 / (s2 + s3) > 50) and the product's pose agrees within 1e-4 with a float64 Kabsch of its OWN correspondences
    and weights (so the Procrustes kernel itself is right); any other failure says 'mismatch'.  Either way the run reports ok: false
    and exits non-zero in the float32 modes."""
    from oracle import canonical, native, regtr_ref
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    torch.set_num_threads(max(1, min(usable_cores(), 16)))
    worst = {'pose_max_abs': 0.0, 'corr_max_abs': 0.0, 'overlap_logit_max_abs': 0.0, 'pose_vs_f64_kabsch_of_own_outputs': 0.0,
             'oracle_f32_vs_f64_kabsch': 0.0, 'kabsch_cond_max': 0.0}
    kp_exact, conditioning_only = True, True
    per_pair = []
    t0 = time.perf_counter()
    for b in which:
        s, t = pairs[b]
        with torch.no_grad():
            if parity_mode:
                ref = regtr_ref.regtr_forward(sd, cfg, [s], [t], use_ref_cpp=True)
            else:
                ref = regtr_ref.regtr_forward(sd, cfg, [s], [t], meta=canonical.canonical_meta([s, t], cfg))
        kp_exact = kp_exact and torch.equal(out['src_kp'][b].cpu(), ref['src_kp'][0]) and torch.equal(out['tgt_kp'][b].cpu(), ref['tgt_kp'][0])
        if not kp_exact:
            break
        e_pose = float((out['pose'][:, b].cpu() - ref['pose'][:, 0]).abs().max())
        e_corr = max(float((out[k][b].cpu() - ref[k][0]).abs().max()) for k in ('src_kp_warped', 'tgt_kp_warped'))
        e_logit = max(float((out[k][b].cpu() - ref[k][0]).abs().max()) for k in ('src_overlap', 'tgt_overlap'))

        def kabsch_inputs(o, i):            # regtr.py:187-194
            L = o['src_kp_warped'][i].shape[0]
            a = torch.cat([o['src_kp'][i].expand(L, -1, -1), o['tgt_kp_warped'][i]], 1).cpu().double()
            bb = torch.cat([o['src_kp_warped'][i], o['tgt_kp'][i].expand(L, -1, -1)], 1).cpu().double()
            w = torch.cat([torch.sigmoid(o['src_overlap'][i][..., 0]), torch.sigmoid(o['tgt_overlap'][i][..., 0])], 1).cpu().double()
            return a, bb, w
        a, bb, w = kabsch_inputs(ref, 0)
        p64_ref = regtr_ref.compute_rigid_transform(a, bb, w)
        wn = w[..., None] / torch.clamp_min(w.sum(-1, keepdim=True)[..., None], 1e-6)
        ca, cb = (a * wn).sum(-2), (bb * wn).sum(-2)
        sv = torch.linalg.svdvals((a - ca[:, None]).transpose(-2, -1) @ ((bb - cb[:, None]) * wn))
        cond = float((sv[:, 0] / (sv[:, 1] + sv[:, 2]).clamp_min(1e-30)).max())
        e_own = float((out['pose'][:, b].cpu().double() - regtr_ref.compute_rigid_transform(*kabsch_inputs(out, b))).abs().max())
        for k, v in (('pose_max_abs', e_pose), ('corr_max_abs', e_corr), ('overlap_logit_max_abs', e_logit), ('kabsch_cond_max', cond),
                     ('pose_vs_f64_kabsch_of_own_outputs', e_own), ('oracle_f32_vs_f64_kabsch', float((ref['pose'][:, 0].double() - p64_ref).abs().max()))):
            worst[k] = max(worst[k], v)
        per_pair.append({'slot': int(b), 'points': [len(s), len(t)], 'pose': e_pose, 'corr': e_corr, 'cond': round(cond, 1)})
        if e_pose >= PARITY_TOL and not (e_corr < PARITY_TOL and cond > 50.0 and e_own < PARITY_TOL):
            conditioning_only = False
    ok = kp_exact and worst['corr_max_abs'] < PARITY_TOL and worst['pose_max_abs'] < PARITY_TOL
    reason = None
    if not ok:
        reason = ('conditioning: correspondences within tol, Kabsch ill-conditioned (s1 / (s2 + s3) > 50) and the pose equals a float64 Kabsch of '
                  'the product\'s own correspondences within tol -- the float32 pose of a near-rank-one problem, not a kernel defect'
                  if (kp_exact and worst['corr_max_abs'] < PARITY_TOL and conditioning_only) else 'mismatch')
    return dict(worst, pairs_checked=len(which), pair_slots=list(which), keypoints_bit_exact=kp_exact, tol=PARITY_TOL, ok=bool(ok), reason=reason,
                pose_gate='direct: R|t within tol of the oracle on every checked pair', per_pair=per_pair,
                vs=('CPU oracle (oracle/regtr_ref.py, pinned to the reference module) per pair, ' +
                    ('reference row / tie orders (oracle/_ref), product in parity mode' if parity_mode else
                     'canonical tables from ' + ('the unmodified reference C++ neighbour sets (oracle/_ref)' if native.have_ref() else 'the C++ restatement'))),
                what='outputs of the last timed step; random-init weights', seconds=round(time.perf_counter() - t0, 2))


def plan_pairs(args, rank, world, device):
    """Which pairs this rank runs and how they are cut into forwards.  Weak scaling (default): `--pairs` pairs per rank and step, ids
    rank * pairs + i.  --config lomatch (configs[3], strong scaling): a fixed set of --total-pairs pairs, pair i -> rank i % world
    (regtr_amd/distributed.py: shard_pairs), cut into the fewest forwards of at most `--pairs` pairs and then into EQUAL ones (a 223-pair
    shard at 192 per forward is 112 + 111, not 192 + 31: a short last forward runs at a fraction of the per-pair rate).
    --emulate-rank-of W: this ONE GPU runs rank 0's shard of a W-rank job (pair i with i % W == 0) -- the per-rank time of a strong-scaling
    run without the node; the line reports it as a PREDICTION.
    -> (lomatch, per_fwd, pair_ids (n_local,) i32 on `device`, chunks [(lo, hi)], pairs_per_step over all ranks)"""
    from regtr_amd.distributed import shard_pairs
    lomatch = args.config == 'lomatch'
    per_fwd = args.pairs if args.pairs else DEFAULT_PAIRS[args.config]
    args.pairs = per_fwd
    emu = getattr(args, 'emulate_rank_of', 0)
    if lomatch:
        mine = shard_pairs(args.total_pairs, 0, emu) if emu else shard_pairs(args.total_pairs, rank, world)
        pair_ids = torch.tensor(mine, device=device, dtype=torch.int32)
    else:
        R = max(1, getattr(args, 'replicas', 1) or 1)          # R concurrent forwards of per_fwd pairs each (workload.ReplicaRunner)
        pair_ids = torch.arange(R * per_fwd, device=device, dtype=torch.int32) + rank * R * per_fwd
    n_local = int(pair_ids.numel())
    n_fwd = max(1, -(-n_local // per_fwd))
    base, extra = divmod(n_local, n_fwd)
    chunks, lo = [], 0
    for i in range(n_fwd):
        hi = lo + base + (1 if i < extra else 0)
        chunks.append((lo, hi)); lo = hi
    return lomatch, per_fwd, pair_ids, chunks, (n_local if emu else (args.total_pairs if lomatch else n_local * world))


def settle_device(step, budget_s, sync=None):
    """Set-up, before the W warm-up steps: `budget_s` seconds of untimed passes, so that the device is at its sustained state when the warm-up
    starts.  Why: on 3 of ~15 fresh boxes the FIRST seconds of sustained GPU work ran 20-45 % slow -- the default line's 20 timed steps 95.7 ms
    each while every side measurement taken a few seconds later in the same process (gather, GEMM and pyramid event timings, the fp32x3 forwards)
    was within 4 % of a normal box; two consecutive processes slow, the following ones not (profiles/r05_z_dist_stream.txt) -- whatever the
    code version and launch mode.  A plateau of slow passes looks steady, so the phase has a fixed length instead of a convergence test;
    `config.settle` reports the first and last pass times.  -> {'passes', 'seconds', 'first_ms', 'last_ms'}"""
    sync = sync or torch.cuda.synchronize
    ts = []
    t_begin = time.perf_counter()
    while time.perf_counter() - t_begin < budget_s:
        sync(); t0 = time.perf_counter()
        step()
        sync(); ts.append((time.perf_counter() - t0) * 1e3)
    return {'passes': len(ts), 'seconds': round(time.perf_counter() - t_begin, 2), 'first_ms': [round(t, 2) for t in ts[:3]], 'last_ms': [round(t, 2) for t in ts[-3:]]}


def timed_passes(args, dist, lomatch, pair_ids, step, sync, device):
    """W untimed + exactly K timed steps between barrier + device synchronisation on both sides, MAX over ranks.  The poses reach every
    rank through ONE all_gather (regtr_amd/distributed.py) -- per pass over the set for lomatch, once at the end of the timed region
    otherwise.  step() -> (poses (n_local, 3, 4) of this rank's pairs, anything).  -> (elapsed s, all poses, all ids, last step's extra)"""
    from regtr_amd.distributed import gather_poses
    n_set = int(pair_ids.numel()) if getattr(args, 'emulate_rank_of', 0) else (args.total_pairs if lomatch else int(pair_ids.numel()) * (dist.get_world_size() if dist else 1))      # (weak scaling: every rank holds the same count)
    gather = (lambda p: gather_poses(p.reshape(-1, 12), pair_ids, n_set)) if dist else (lambda p: (p.reshape(-1, 12), pair_ids))
    warm = None
    for _ in range(args.warmup):
        warm, _ = step()
    if dist and warm is not None:
        # a warm-up pass ends like a timed one, with the pose gather: the FIRST all_gather_into_tensor of a process builds RCCL's channels and
        # loads its kernels -- ~80 ms measured at world 1 (round 5 A/B, profiles/r05_z_dist_stream.txt: 87.7 vs 78.0 ms per step over 8 steps), which a plain
        # `python bench.py` (no process group) never pays and a torchrun launch would otherwise pay inside the timed region
        gather(warm)
    sync()
    if dist: dist.barrier()
    t0 = time.perf_counter()
    poses = extra = gathered = None
    if not lomatch and hasattr(step, 'many'):
        poses, extra = step.many(args.steps)       # R host threads run their K forwards each without meeting in between (workload.ReplicaRunner)
    else:
        for _ in range(args.steps):
            poses, extra = step()
            if lomatch:
                gathered = gather(poses)
    if not lomatch:
        gathered = gather(poses)
    sync()
    if dist: dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist:       # (outside the timed region) every rank's own clock, so a straggler is visible; the reported time is the MAX
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = torch.empty(dist.get_world_size(), device=device, dtype=torch.float64)
        dist.all_gather_into_tensor(every, tt)
        per_rank = [float(v) for v in every.cpu()]
        elapsed = max(per_rank)
    return elapsed, gathered[0], gathered[1], extra, per_rank


def count_ranks(all_ids, lomatch, per_rank, world):
    """Ranks whose poses arrived through the gather (the owner of pair id i is i % world in a sharded set, i // (pairs per rank) otherwise)."""
    owner = torch.remainder(all_ids, world) if lomatch else torch.div(all_ids, per_rank, rounding_mode='floor')
    return int(torch.unique(owner).numel())


def run_stub(args, rank, world, dist):
    """tests/test_bench_entry.py: the launch / sharding / gather / reporting logic of this script on CPU (gloo) with a
    stand-in for the forward -- no kernels, no claims; prints the same JSON shape with metric 'stub'."""
    cpu = torch.device('cpu')
    args.replicas = 1
    lomatch, per_fwd, pair_ids, chunks, pairs_per_step = plan_pairs(args, rank, world, cpu)
    eye = torch.eye(3, 4).reshape(1, 3, 4)

    def step():
        return torch.cat([eye + pair_ids[lo:hi, None, None].float() for lo, hi in chunks]), None
    elapsed, all_poses, all_ids, _, per_rank = timed_passes(args, dist, lomatch, pair_ids, step, lambda: None, cpu)
    ranks_seen = count_ranks(all_ids, lomatch, int(pair_ids.numel()), world)
    assert all_poses.shape[0] == pairs_per_step and ranks_seen == world
    assert torch.equal(all_poses[:, 0], 1 + all_ids.float())
    assert torch.equal(all_ids, torch.sort(all_ids)[0]) and (not lomatch or torch.equal(all_ids.long(), torch.arange(args.total_pairs)))
    if rank == 0:
        print(json.dumps({'metric': 'stub', 'value': args.steps * pairs_per_step / max(elapsed, 1e-9), 'unit': 'pairs/s',
                          'n_gpus': ranks_seen, 'steps': args.steps, 'warmup': args.warmup, 'pairs_per_step': pairs_per_step,
                          'forwards_per_step_rank0': len(chunks), 'scaling': 'strong' if lomatch else 'weak',
                          'per_rank_ms_per_step': [t / args.steps * 1e3 for t in per_rank]}))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', choices=['3dmatch', 'modelnet', 'lomatch'], default='3dmatch',
                    help='3dmatch = BASELINE configs[2] (the headline metric); modelnet = configs[1] (ModelNet-size pairs, bf16 cross-encoder); '
                         'lomatch = configs[3]: --total-pairs 10-30 %%-overlap pairs sharded over the ranks (strong scaling), a step = one pass over the set')
    ap.add_argument('--total-pairs', type=int, default=1781, help='lomatch: size of the pair set (3DLoMatch test list: 1781)')
    ap.add_argument('--emulate-rank-of', type=int, default=0, metavar='W',
                    help='lomatch: run rank 0\'s shard of a W-rank job (pairs i %% W == 0) on this one GPU, pose gather through a one-rank RCCL group included, and report '
                         'predicted_pairs_per_s = total pairs / shard time -- a strong-scaling PREDICTION for W GPUs, flagged as such (no multi-GPU node needed)')
    ap.add_argument('--distinct-pairs', type=int, default=128, help='lomatch: different synthetic pairs generated per rank (cycled; set-up time only)')
    ap.add_argument('--parity-mode', action='store_true', help='cfg.kpconv_ref_row_order: the reference CPU ops\' row / tie orders on the GPU (slower; DESIGN section 4)')
    ap.add_argument('--parity-pairs', type=int, default=8, help='pairs of the last timed step checked against the CPU oracle: first, last, largest, smallest slot of the batch + evenly spaced others (0 = off)')
    ap.add_argument('--dtype', choices=['fp32', 'fp32x3', 'bf16', 'bf16x2'], default=None, help='cfg.compute_dtype (default: fp32 for 3dmatch, bf16 for modelnet)')
    ap.add_argument('--pairs', type=int, default=0, help='pairs per FORWARD (default 64 for 3dmatch -- three forwards in flight, see --replicas: 192 pairs per step --, 128 for modelnet -- three in flight --, at most 64 per forward for lomatch: a shard is cut into equal forwards; pairs are independent, 288 GB of HBM holds far more)')
    ap.add_argument('--points', type=int, default=20000, help='approx. points per cloud')
    ap.add_argument('--shuffle', action='store_true', help='randomly permute the points of every cloud (worst-case gather locality)')
    ap.add_argument('--real', action='store_true', help='3dmatch: the three REAL pairs the reference ships (demo.py:26-49; tests/golden fixtures) replicated to --pairs under random rigid motions, instead of synthetic rooms')
    ap.add_argument('--head-init', choices=['uniform', 'probe'], default=None, help="output layer of the correspondence head: U(-0.5, 0.5) (3dmatch default) or a linear probe for the tokens' coordinates (modelnet default): bench.probe_head")
    ap.add_argument('--settle-s', type=float, default=5.0, help='set-up: seconds of untimed passes before the warm-up steps (the first seconds of sustained GPU work are slow on some boxes; 0 = off)')
    ap.add_argument('--replicas', type=int, default=0,
                    help='concurrent forwards per GPU: R model replicas on R host threads / HIP streams, each running its own --pairs-pair forwards (a step = R '
                         'forwards = R x --pairs pairs).  Default 3 for the 3dmatch / lomatch configurations (3 x 64 pairs: +6 % over one 192-pair forward at a time -- '
                         'forwards in flight fill each other\'s host waits and 5-6 ms heads; workload.ReplicaRunner), 3 for modelnet (3 x 128), 1 in parity mode and for --pairs < 16 of ~20 k-point clouds; '
                         '`--pairs 192 --replicas 1` = the round-5 line')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-real', action='store_true', help='skip the side measurement of the same configuration on the shipped real fragments (real_fragments_pairs_per_s)')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-strict-f32', action='store_true', help="skip the side measurement of compute_dtype 'fp32x3' on the same workload")
    ap.add_argument('--no-range-check', action='store_true', help='diagnostic: cfg.f16_range_check off -- no status-word wait at the end of a forward, so consecutive forwards are enqueued back to back')
    ap.add_argument('--stub-backend', default=None, help=argparse.SUPPRESS)   # tests/test_bench_entry.py: 'gloo'
    ap.add_argument('--collect-pmc', action='store_true', help='run the rocprofv3 counter passes behind roofline.traffic on this workload and write profiles/pmc_traffic.json (stamped with the code version)')
    ap.add_argument('--pmc-tag', default='r05', help=argparse.SUPPRESS)
    ap.add_argument('--cpu-baseline-only', action='store_true',
                    help='no GPU needed: time only the CPU baseline leg on the synthetic workload and print it (where /root/reference '
                         'exists this times the REAL reference module, kind "reference")')
    args = ap.parse_args()
    if args.collect_pmc:
        return collect_pmc(args)
    if args.cpu_baseline_only:
        from regtr_amd.config import load_config
        arch = '3dmatch' if args.config == 'lomatch' else args.config
        cfg = load_config(os.path.join(ROOT, 'regtr_amd', 'conf', f'{arch}.yaml'))
        gen = ((lambda i: synth_modelnet_pair(i)) if args.config == 'modelnet' else
               (lambda i: synth_pair(i, args.points, args.shuffle, overlap='lomatch' if args.config == 'lomatch' else None)))
        print(json.dumps({'cpu_baseline': cpu_baseline(cfg, [gen(i) for i in range(24 if args.config == 'modelnet' else 6)],
                                                       cfg_name=arch), 'config': args.config}))
        return

    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become N ranks (one process per GPU) under torch.distributed.run
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-GPU run as {args.gpus} GPUs')
    stub = args.stub_backend is not None      # tests only: the multi-process entry logic on CPU (gloo), no kernels
    dist = None
    if args.emulate_rank_of:
        if args.config != 'lomatch' or args.gpus != 1 or args.emulate_rank_of < 1:
            ap.error('--emulate-rank-of W needs --config lomatch on one GPU')
        if 'RANK' not in os.environ:              # a one-rank RCCL group of our own, so that the shard's pose gather is the real collective
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ.update(RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(sk.getsockname()[1]))
    if world > 1 or 'RANK' in os.environ:        # launched by torch.distributed.run: a process group even at one rank (RCCL init, the
        import torch.distributed as dist        # device-tensor all_gather) -- the same code path at every world size
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if stub:
            dist.init_process_group(args.stub_backend)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    if stub:
        return run_stub(args, rank, world, dist)
    if not torch.cuda.is_available() or local_rank >= torch.cuda.device_count():
        sys.exit(f'bench.py: rank {rank} needs GPU {local_rank}; {torch.cuda.device_count()} visible (there is no CPU path)')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    dtype = args.dtype or ('bf16' if args.config == 'modelnet' else 'fp32')
    if args.replicas < 1:
        # (small forwards of ~20 k-point clouds -- --pairs 1 / 3 / 8: the reference's loop -- stay one at a time by default: their lines quote ms per FORWARD;
        #  three in flight: 688 / 1377 / 1720 pairs/s, profiles/r06_w_*)
        args.replicas = 1 if (args.parity_mode or (args.pairs and args.pairs < 16 and args.points < 50000)) else DEFAULT_REPLICAS[args.config]
    lomatch, per_fwd, pair_ids, chunks, pairs_per_step = plan_pairs(args, rank, world, dev)
    n_local = int(pair_ids.numel())
    if args.real and args.config != '3dmatch':
        sys.exit('bench.py: --real is the 3dmatch configuration on the shipped real fragments')
    cfg, model, pairs, batch = build_workload(args.config, n_local, args.points, args.shuffle, rank, dev, dtype, args.parity_mode,
                                              distinct=args.distinct_pairs if lomatch else None, real=args.real, head_init=args.head_init)
    if args.no_range_check:
        model._range_check = False

    n_rep = min(args.replicas, len(chunks))
    runner = ReplicaRunner(replicate(model, cfg, n_rep, dev), batch, chunks, dev)

    def step():
        outs = runner.run(1)
        return runner.poses(outs), outs[-1]

    def many(n):
        outs = runner.run(n)
        return runner.poses(outs), outs[-1]
    if n_rep > 1:
        step.many = many
    settle = settle_device(step, args.settle_s)
    elapsed, all_poses, all_ids, last_out, per_rank = timed_passes(args, dist, lomatch, pair_ids, step, torch.cuda.synchronize, dev)
    peak_gb = torch.cuda.max_memory_allocated(dev) / 2**30       # inputs, weights and every buffer of the forwards so far
    assert all_poses.shape[0] == pairs_per_step and torch.isfinite(all_poses).all()
    ranks_seen = count_ranks(all_ids, lomatch, n_local, world)                                      # who entered the all_gather
    assert ranks_seen == world, (ranks_seen, world)

    if rank == 0:
        total_pairs = args.steps * pairs_per_step
        mean_pts = [int(np.mean([len(s) for s, _ in pairs])), int(np.mean([len(t) for _, t in pairs]))]
        if args.config == 'modelnet':
            metric = f'point-cloud pairs/sec (ModelNet ~{mean_pts[0]} pts)'
            workload = 'BASELINE configs[1]: ModelNet40-benchmark-size pairs, 6-layer cross-attn in ' + dtype
        else:
            metric = 'point-cloud pairs/sec (3DMatch ~20k pts)'
            workload = 'BASELINE configs[2]: 3DMatch-size pairs, full KPConv encoder + 6-layer cross-attn + SVD'
            if args.points != 20000:
                workload = f'BASELINE configs[4]-style stress: ~{args.points}-point clouds, conf/3dmatch.yaml pipeline'
            if lomatch:
                metric = 'point-cloud pairs/sec (3DLoMatch-like set, ~20k pts, overlap 10-30 %)'
                workload = (f'BASELINE configs[3]: {args.total_pairs} 3DLoMatch-like pairs (overlap 10-30 %) sharded pair i -> rank i % {args.emulate_rank_of or world}, '
                            f'forwards of {[hi - lo for lo, hi in chunks]} pairs on rank 0, one RCCL pose all_gather per pass; a step = one pass over the ' + ('SHARD of rank 0 (emulation) ' if args.emulate_rank_of else 'set ') +
                            f'({min(args.distinct_pairs, n_local)} distinct synthetic pairs per rank, cycled)')
            if args.real:
                metric = 'point-cloud pairs/sec (3DMatch REAL fragments, ~17-25k pts)'
                workload = ('BASELINE configs[2] on REAL data: the three 3DMatch pairs the reference ships (demo.py:26-49: red-kitchen, hotel_umd, '
                            f'home_at) replicated to {args.pairs} per forward, every replica under its own random rigid motions (rot <= 45 deg, |t| <= 0.5 m); '
                            'full KPConv encoder + 6-layer cross-attn + SVD')
            if args.parity_mode:
                workload += ' [PARITY MODE: reference row / tie orders reproduced on the GPU]'
        if n_rep > 1 and not lomatch:
            workload += f'; {n_local} pairs per step as {n_rep} forwards of {per_fwd} pairs in flight (model replicas on host threads / HIP streams)'
        res = {
            'metric': metric, 'value': total_pairs / elapsed, 'unit': 'pairs/s',
            'n_gpus': ranks_seen, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3,
            'per_rank_ms_per_step': [round(t / args.steps * 1e3, 3) for t in per_rank],
            'higher_is_better': True, 'scaling': 'strong' if lomatch else 'weak', 'vs_baseline': None,
            'dtype': {'fp32': 'f32 (f16-pair split, 22-bit operands)', 'fp32x3': 'f32 (bf16x3 split, 24-bit operands)'}.get(dtype, dtype),
            'data': 'real 3DMatch fragments (3 shipped pairs, replicated under random rigid motions)' if args.real else 'synthetic',
            'config': {'workload': workload, 'pairs_per_step_per_gpu': n_local, 'pairs_per_forward': [hi - lo for lo, hi in chunks], 'concurrent_forwards': n_rep,
                       'concurrency': (f'{n_rep} model replicas on {n_rep} host threads / HIP streams, each running its own forwards (workload.ReplicaRunner); timed steps free-running' if n_rep > 1 else 'one forward at a time'),
                       'points_per_cloud': mean_pts,
                       'arch': f'conf/{"3dmatch" if lomatch else args.config}.yaml, random-init weights; head output layer: ' + ('U(-0.5, 0.5) (predictions spread over metres: a well-conditioned Procrustes problem)' if model.head_init == 'uniform' else f"linear probe for the tokens' own coordinates fitted on 4 calibration pairs (r^2 {model.head_probe_r2:.2f}; bench.probe_head: correspondences correlated with the key points, as a trained head's are -- a well-conditioned Procrustes problem)"), 'compute_dtype': dtype, 'shuffle': bool(args.shuffle),
                       'parallelism': f'pair-sharded x{world}, ONE RCCL all_gather_into_tensor of the (pose | id) rows', 'peak_hbm_allocated_GiB': round(peak_gb, 2),
                       'settle': dict(settle, what='untimed set-up passes before the warm-up steps (bench.settle_device)'),
                       'arithmetic': {'fp32': 'float32-grade: exact operand splits on the 16-bit matrix cores (f16 pair, three MFMA terms, where the strip GEMM / attention '
                                              'kernels serve the shape; bf16x3, six terms, elsewhere), float32 accumulation; exact-f32 MFMA in the KPConv gather',
                                      'fp32x3': 'float32-grade: bf16x3 operand splits (six MFMA terms) everywhere, float32 accumulation',
                                      'bf16x2': 'bf16 three-term split in the cross-encoder Linears, bf16x3 elsewhere',
                                      'bf16': 'plain bf16 operands in the cross-encoder Linears and attention core, float32-grade elsewhere'}[dtype]},
        }
        if args.emulate_rank_of:
            W = args.emulate_rank_of
            res['predicted'] = {
                'world': W, 'predicted_pairs_per_s': args.total_pairs / (elapsed / args.steps), 'shard_pairs': n_local, 'shard_ms': elapsed / args.steps * 1e3,
                'forwards_per_shard': [hi - lo for lo, hi in chunks],
                'PREDICTION': f'rank 0\'s shard of a {W}-rank pass over the {args.total_pairs}-pair set, timed on ONE GPU with the pose gather through a one-rank RCCL group: '
                              f'a {W}-GPU pass takes at least this long (largest shard; the {W}-rank all_gather moves {W} x {-(-args.total_pairs // W)} x 52 bytes over xGMI instead of one '
                              'rank\'s rows, and the ranks share the host).  NOT a measurement of a multi-GPU run.'}
        if args.parity_pairs > 0:
            # "pose err vs ref" (BASELINE.json metric): the last timed forward's outputs against the CPU oracle, >= 2 pairs
            lo, hi = chunks[-1]
            slots = parity_slots([len(a) + len(b) for a, b in pairs[lo:hi]], args.parity_pairs)
            res['parity'] = parity_check(cfg, model, pairs[lo:hi], last_out, slots, args.parity_mode)
            res['parity']['enforced'] = dtype in ('fp32', 'fp32x3')      # 'bf16' reports the error; the 1e-4 gate is the float32 modes'
        fwd_batch = {k: v[chunks[0][0]:chunks[0][1]] for k, v in batch.items()}
        if not args.no_roofline:
            keep = {}
            r = measure_kpconv_roofline(model, fwd_batch, keep=keep)
            traffic = pmc_traffic(args.pairs, args.points, args.shuffle, r, args.real) if (args.config == '3dmatch' and not args.parity_mode) else None
            gather = {'bound': 'hbm', 'achieved': r['achieved_gather_kernel_GBs'], 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                      'frac': r['achieved_gather_kernel_GBs'] / HBM_PEAK_GBS, 'traffic': traffic, 'detail': r}
            if gather['frac'] > 1.0:      # small clouds (ModelNet-size): the gathered rows never leave L2 / Infinity Cache
                gather['note'] = ('algorithmic bytes exceed what HBM can deliver: at this size the feature rows are cache-resident, so this is '
                                  'not an HBM-bound launch and the fraction is not a roofline fraction')
            a = measure_attention(model, fwd_batch, cfg.nhead, cfg.d_embed, cfg.num_encoder_layers)
            peak = MFMA_BF16_PEAK_TFS
            att = {'bound': 'mfma', 'achieved': a['achieved_TFs'], 'peak': peak, 'unit': 'TFLOP/s', 'frac': a['achieved_TFs'] / peak,
                   'traffic': None, 'detail': dict(a, operands={'fp32': 'f16 pair split (3 MFMAs per product, float32-grade)' if model.transformer_encoder.layers[0].attn_precision == 3 else 'bf16x3 split (6 MFMAs per product, float32-grade)',
                                                               'fp32x3': 'bf16x3 split (6 MFMAs per product, float32-grade)', 'bf16x2': 'bf16x3 split',
                                                               'bf16': 'plain bf16 (1 MFMA per product)'}[dtype],
                                                   peak_note='dense bf16 / f16 MFMA peak; the split modes issue 6x (bf16x3) / 3x (f16 pair) the algorithmic flops')}
            # the dominant kernel of the configuration leads: KPConv gather (HBM) for 3DMatch-size pairs, attention (MFMA) for ModelNet
            res['config']['code'] = code_version()
            res['roofline'] = att if args.config == 'modelnet' else gather
            res['roofline_secondary'] = gather if args.config == 'modelnet' else att
            res['roofline_gemm'] = measure_gemm_roofline(model, fwd_batch, keep=keep)
            res['preprocess'] = measure_preprocess(model, fwd_batch)
            # the WHOLE forward against its own roofline (round 6): compulsory bytes / 8 TB/s and the matrix-pipe time of every product at the
            # peaks; the step can be no faster than the larger of the two.  And the HBM bytes the counters saw the forward move, next to them.
            from regtr_amd.workload import forward_compulsory_bytes, forward_matrix_seconds
            lv = res['preprocess']['level_points']
            comp = forward_compulsory_bytes(model, lv, list(cfg.neighborhood_limits), lv[-1], cfg.d_embed)
            terms_att = {3: 3}.get(model.transformer_encoder.layers[0].attn_precision, 6) if dtype in ('fp32', 'fp32x3') else (1 if dtype == 'bf16' else 6)
            fm = forward_matrix_seconds(keep['gemm'][0], keep['gemm'][1], keep['gather'][0], keep['gather'][1], a['alg_flops_per_step'], terms_att)
            t_matrix = fm['dense_s'] + fm['gather_s'] + fm['attention_s']
            t_hbm = comp / (HBM_PEAK_GBS * 1e9)
            fwd_ms = elapsed / args.steps / len(chunks) * 1e3
            res['forward_roofline_frac'] = round(max(t_hbm, t_matrix) * 1e3 / fwd_ms, 4)
            res['forward_roofline'] = {
                'what': 'max(compulsory HBM bytes / 8 TB/s, matrix-pipe time of every product at the peaks) / measured time of one forward',
                'forward_ms': round(fwd_ms, 3), 'hbm_ms': round(t_hbm * 1e3, 3), 'matrix_ms': round(t_matrix * 1e3, 3),
                'matrix_ms_split': {'dense_products': round(fm['dense_s'] * 1e3, 3), 'kpconv_gather_correlation_f32_mfma': round(fm['gather_s'] * 1e3, 3), 'attention_core': round(fm['attention_s'] * 1e3, 3)},
                'algorithmic_TFLOP': round(fm['algorithmic_flops'] / 1e12, 3), 'issued_TFLOP': round(fm['issued_flops'] / 1e12, 3),
                'peaks': {'hbm_GBs': HBM_PEAK_GBS, 'mfma_16bit_dense_TFLOPs': MFMA_BF16_PEAK_TFS, 'mfma_f32_TFLOPs': 157.3},
                'kernel_time_sum_ms': {'dense_products': res['roofline_gemm']['ms_per_step'], 'kpconv_gathers': round(r['gather_s_per_step'] * 1e3, 3),
                                       'attention_core': round(a['attention_s_per_step'] * 1e3, 3), 'pyramid_alone': res['preprocess']['pyramid_ms_alone'],
                                       'note': 'each family event-timed alone on ONE stream; their sum exceeds the step by what the two-stream forward overlaps'}}
            res['forward_traffic'] = forward_traffic(args.pairs, args.points, args.shuffle, args.real, comp) if args.config == '3dmatch' and not args.parity_mode else \
                {'hbm_GB': None, 'compulsory_GB': round(comp / 1e9, 3), 'ratio': None, 'top3': None, 'note': 'counter passes are kept for the default 3dmatch workload only'}
            # the candid companion of roofline.frac, at the top level: HBM bytes the COUNTERS saw per gather launch / launch time / peak
            res['counter_hbm_frac_of_peak'] = r.get('counter_hbm_frac_of_peak')
        if dtype not in ('fp32', 'fp32x3'):
            # reduced-precision error, reported next to the number (parity is gated in fp32): same batch, float32-grade model
            from regtr_amd import RegTR, load_config
            cfg32 = load_config(os.path.join(ROOT, 'regtr_amd', 'conf', f'{"3dmatch" if lomatch else args.config}.yaml'))
            m32 = RegTR(cfg32).to(dev).eval()
            m32.load_state_dict(model.state_dict())
            sub = {'src_xyz': list(fwd_batch['src_xyz'][:16]), 'tgt_xyz': list(fwd_batch['tgt_xyz'][:16])}
            o32 = m32(dict(sub)); olo = model(dict(sub))
            nb = len(sub['src_xyz'])
            res['reduced_precision_error'] = {
                'vs': 'float32-grade run of the same weights / pairs', 'pairs': nb,
                'max_abs_correspondence': max(float((olo['src_kp_warped'][b] - o32['src_kp_warped'][b]).abs().max()) for b in range(nb)),
                'max_abs_pose': float((olo['pose'] - o32['pose']).abs().max())}
            # the reduced-precision line's own gate (enforced: non-zero exit).  bf16 operands carry 8 bits, so the float32 modes' 1e-4 bar cannot
            # apply; measured on this workload: correspondences 5.5e-3 ... 6.0e-3, pose 1.5e-2 ... 2.9e-2 (unit-scale objects).  The bounds sit
            # ~3x above that: they do not certify bf16, they catch a broken reduced-precision path (errors of order 1).
            rp = res['reduced_precision_error']
            rp['gate'] = {'tol_correspondence': REDUCED_TOL['correspondence'], 'tol_pose': REDUCED_TOL['pose'], 'enforced': True,
                          'ok': bool(rp['max_abs_correspondence'] <= REDUCED_TOL['correspondence'] and rp['max_abs_pose'] <= REDUCED_TOL['pose'])}
            if 'parity' in res:
                res['parity']['note'] = ("the 1e-4 bar is the float32 modes' (`enforced`: false here); this line is gated by reduced_precision_error.gate "
                                         "against the float32-grade run of the same weights and pairs")
        def side(name, fn):
            """A side measurement outside the timed region: its failure is recorded in the line (`<name>_error`), it does not cost the headline."""
            try:
                fn()
            except Exception as e:      # noqa: BLE001
                res[name + '_error'] = f'{type(e).__name__}: {e}'[:400]
                torch.cuda.empty_cache()

        def measure_fp32x3():
            # the same workload with strictly 24-bit operands (compute_dtype 'fp32x3': six-term bf16 splits everywhere), quoted beside the
            # default line whose dense operands carry 22 bits: same weights, same batch, measured here, outside the timed region
            from regtr_amd import RegTR
            cfg3 = cfg.copy() if hasattr(cfg, 'copy') else cfg
            cfg3.update({'compute_dtype': 'fp32x3'})
            m3 = RegTR(cfg3).to(dev).eval()
            m3.load_state_dict(model.state_dict())
            run3 = ReplicaRunner(replicate(m3, cfg3, n_rep, dev), batch, chunks, dev)       # the same forwards in flight as the timed line
            cfg.update({'compute_dtype': dtype})                                            # (cfg3 may alias cfg)
            k3 = max(2, min(args.steps, 8))
            run3.run(2)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            o3 = run3.run(k3)
            torch.cuda.synchronize(); t3 = (time.perf_counter() - t3) / k3
            res['fp32x3_pairs_per_s'] = n_local / t3      # strictly 24-bit operands, same weights, batch and concurrency, this run
            res['config']['fp32x3_same_workload'] = {'value': n_local / t3, 'unit': 'pairs/s', 'ms_per_step': t3 * 1e3, 'steps': k3,
                                                     'max_abs_pose_vs_default': float((o3[0]['pose'] - model(dict(fwd_batch))['pose']).abs().max())}
            del run3, o3, m3
        if dtype == 'fp32' and world == 1 and not args.no_strict_f32 and not args.parity_mode:
            side('fp32x3', measure_fp32x3)
        if args.config == '3dmatch' and world == 1 and not args.real and not args.no_real and not args.parity_mode and args.points == 20000 and not args.shuffle:
            # the same configuration on the REAL fragments the reference ships (demo.py:26-49), same pairs per forward, measured in this run outside
            # the timed region: the synthetic rooms are calibrated to the red-kitchen pair's level sizes (regtr_amd/synthetic.py), this is the check
            def measure_real():
                res['real_fragments_pairs_per_s'], res['real_fragments'] = measure_real_fragments(args, dev, dtype)
            side('real_fragments', measure_real)
        if not args.no_cpu_baseline and world == 1:      # the CPU leg is reported by the single-GPU run only
            try:
                res['cpu_baseline'] = cpu_baseline(cfg, [pairs[i % len(pairs)] for i in range(24 if args.config == 'modelnet' else 6)],
                                                   cfg_name='3dmatch' if lomatch else args.config)
            except Exception as e:      # noqa: BLE001  (the CPU leg must not cost the GPU line)
                res['cpu_baseline'] = {'value': None, 'unit': 'pairs/s', 'cores': 0, 'kind': 'port', 'sample': f'FAILED: {type(e).__name__}: {e}'[:300]}
            # the REAL reference module cannot run on a GPU box (no /root/reference there): its rate on the same synthetic workload, measured
            # in the build container (`python bench.py --cpu-baseline-only`), travels as a committed profile and is quoted beside the port
            try:
                # (3dmatch: re-measured in round 6 on the calibrated synthetic pairs; the ModelNet-size generator has not changed since round 2)
                ref_file = os.path.join(ROOT, 'profiles', 'r02_cpu_baseline_reference_modelnet.json' if args.config == 'modelnet' else 'r06_cpu_baseline_reference_3dmatch.json')
                rb = json.load(open(ref_file))['cpu_baseline']
                if res['cpu_baseline'].get('value') is not None:
                  res['cpu_baseline']['reference_module_build_container'] = {
                    'value': rb['value'], 'unit': rb['unit'], 'cores': rb['cores'], 'kind': rb['kind'], 'sample': rb['sample'],
                    'source': os.path.relpath(ref_file, ROOT), 'note': 'a different host (the 8-core build container), not this box: never to be '
                    'compared with `value` as a speed-up of one over the other'}
            except (OSError, KeyError, ValueError):
                pass
        # the JSON line is the LAST line of stdout: RCCL writes its version banner through C stdio, which (into a pipe or a file) is flushed at
        # exit and would land behind a Python print -- push it out first
        try:
            import ctypes
            sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and res.get('parity', {}).get('enforced') and not res.get('real_fragments', {}).get('parity', {}).get('ok', True):
        sys.exit(f"bench.py: PARITY FAILED on the real-fragment batch -- {res['real_fragments']['parity']}")
    if rank == 0 and res.get('parity', {}).get('enforced') and not res['parity']['ok']:
        sys.exit(f"bench.py: PARITY FAILED -- {res['parity']}")
    if rank == 0 and not res.get('reduced_precision_error', {}).get('gate', {}).get('ok', True):
        sys.exit(f"bench.py: REDUCED-PRECISION GATE FAILED -- {res['reduced_precision_error']}")


if __name__ == '__main__':
    main()
