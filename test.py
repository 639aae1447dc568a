"""Evaluation entry point with the reference's command line (/root/reference/src/test.py:13-29):

    python test.py --benchmark {3DMatch,3DLoMatch,ModelNet,ModelLoNet} [--config CFG] [--resume CKPT]
                   [--logdir DIR] [--dev] [--name NAME] [--num_workers N]
    (N GPUs: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 test.py ...)

Runs the MI355X-native RegTR inference path over the benchmark's pairs and writes `<log>/<benchmark>/<scene>/est.log`
(3DMatch / 3DLoMatch) or `<log>/pred_transforms.npy` (ModelNet / ModelLoNet) in the reference's formats
(models/generic_reg_model.py:194-195, 260-281), which the reference's evaluation scripts read unchanged.
Inference only: no loss, no tensorboard.  Extra flags: --batch (pairs per forward), --data_root, --synthetic N
(N deterministic synthetic pairs instead of the dataset files), --max_pairs, --preprocessor (cpu | gpu: which reference preprocessor's semantics;
--neighbor_order / --voxel_key set its two rules one by one), --benchmark_dir (gt.log / gt.info folder: the
Predator registration-recall table of benchmark/benchmark_predator.py is then printed, computed in process).
"""
import argparse
import logging
import os
import sys
import time

os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')      # idle OpenMP workers must not spin against a cgroup CPU quota

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

parser = argparse.ArgumentParser()
parser.add_argument('--benchmark', type=str, help='Benchmark dataset', default='3DMatch',
                    choices=['3DMatch', '3DLoMatch', 'ModelNet', 'ModelLoNet'])
parser.add_argument('--config', type=str, help='Path to the config file.')
parser.add_argument('--logdir', type=str, default='../logs', help='Directory to store logs, summaries, checkpoints.')
parser.add_argument('--dev', action='store_true', help='If true, will ignore logdir and log to ../logdev instead')
parser.add_argument('--name', type=str, help='Prefix to add to logging directory')
parser.add_argument('--num_workers', type=int, default=-1,
                    help='loader PROCESSES (reference test.py:26; there one pair per step, here each assembles whole batches into pinned slabs: '
                         'regtr_amd/harness.py LoaderPool); 0 = one loader thread in this process; default -1 = 4 (more than ~6 compete with the launching '
                         'thread for a 16-core quota and slow the set down)')
parser.add_argument('--resume', type=str, help='Checkpoint to resume from')
# harness options (not in the reference)
parser.add_argument('--batch', type=int, default=64,
                    help='pairs per forward (64: end to end over the 1781-pair set 1762 pairs/s against 1668 at 192 -- ten forwards do not amortise the pipeline fill; bench.py measures 192 on resident inputs; 1 = the reference loop)')
parser.add_argument('--replicas', type=int, default=0,
                    help='forwards in flight: R model replicas (same weights) on R host threads / HIP streams taking batches off the one loader in turn '
                         '(regtr_amd/harness.py run_test; default 3)')
parser.add_argument('--data_root', type=str, default=None, help='overrides cfg.root (folder holding test/<scene>/cloud_bin_*.pth)')
parser.add_argument('--info', type=str, default=None, help='benchmark info pickle (default: datasets/3dmatch/test_<benchmark>_info.pkl)')
parser.add_argument('--synthetic', type=int, default=0, help='run N synthetic pairs instead of the dataset files')
parser.add_argument('--materialize', type=str, default=None,
                    help='with --synthetic N: first write the N pairs as <DIR>/test/<scene>/cloud_bin_*.pth + an info pickle (the 3DMatch test-set '
                         'layout, data_loaders/threedmatch.py:74-75) and run the DATASET path over them: .pth loads on the worker thread, pinned '
                         'H2D copies, forwards, pose gather, est.log writes -- the end-to-end figure of the harness')
parser.add_argument('--distinct', type=int, default=0, help='with --materialize: generate only this many different pairs (pair i = pair i %% distinct; one file pair per pair all the same)')
parser.add_argument('--overlap', type=str, default=None, help="synthetic pairs: 'lomatch' = 10-30 %% overlap (3DLoMatch-like)")
parser.add_argument('--max_pairs', type=int, default=None)
parser.add_argument('--warmup_points', type=int, default=24000, help='points per cloud of the warm-up batch (larger than the data so that later batches fit the allocator\'s blocks)')
parser.add_argument('--alloc_conf', type=str, default='', help='torch caching-allocator settings for the run (e.g. roundup_power2_divisions:8)')
parser.add_argument('--reserve_gb', type=float, default=0, help='GiB reserved once and returned to the caching allocator\'s pool before the run (default 0 = off; measured: no effect on the forward-time spikes, which were GIL hand-offs, not hipMalloc)')
parser.add_argument('--no_warmup', action='store_true', help='skip the untimed warm-up forward (weight re-layout, allocator growth) before the timed loop')
parser.add_argument('--cache_dir', type=str, default=None,
                    help='mirror the torch-saved .pth fragments there once as float32 .npy files (np.load: ~0.1 ms against ~2 ms per fragment)')
parser.add_argument('--neighbor_order', choices=('nearest', 'index'), default=None,
                    help='neighbour selection rule: nearest = the reference CPU Preprocessor (default), index = its PreprocessorGPU '
                         '(pytorch3d ball_query: first K supports of a ball by index); overrides cfg.kpconv_neighbor_order')
parser.add_argument('--voxel_key', choices=('origin', 'floor', 'floor_rcp'), default=None,
                    help='voxel rule of the grid subsampling: origin = the reference CPU Preprocessor (default), floor = its PreprocessorGPU '
                         '(MinkowskiEngine: floor(p / dl), no origin shift); overrides cfg.kpconv_voxel_key')
parser.add_argument('--preprocessor', choices=('cpu', 'gpu'), default=None,
                    help="which reference preprocessor's semantics: cpu = Preprocessor (nearest + origin; default, pinned), "
                         'gpu = PreprocessorGPU, the class the reference model instantiates (index + floor)')
parser.add_argument('--benchmark_dir', type=str, default=os.path.join('datasets', '3dmatch', 'benchmarks'),
                    help='folder with <benchmark>/<scene>/gt.log, gt.info for the registration-recall table')


def prepare_logger(opt, rank):
    """cvhelpers.misc.prepare_logger: <logdir>/<yymmdd_HHMMSS>[_name] (or ../logdev with --dev), log.txt inside."""
    if opt.dev:
        log_path = '../logdev'
    else:
        stamp = time.strftime('%y%m%d_%H%M%S')
        log_path = os.path.join(opt.logdir, stamp if not opt.name else f'{stamp}_{opt.name}')
    logger = logging.getLogger('regtr_amd.test')
    logger.setLevel(logging.INFO)
    handlers = [logging.StreamHandler()]
    if rank == 0:
        os.makedirs(log_path, exist_ok=True)
        handlers.append(logging.FileHandler(os.path.join(log_path, 'log.txt')))
    for h in handlers:
        h.setFormatter(logging.Formatter('%(asctime)s [%(levelname)s] %(message)s'))
        logger.addHandler(h)
    return logger, log_path


def main():
    opt = parser.parse_args()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    logger, opt.log_path = prepare_logger(opt, rank)

    # config resolution exactly as test.py:33-52
    if opt.config is None:
        if opt.resume is None or not os.path.exists(opt.resume):
            logger.error('--config needs to be supplied unless resuming from checkpoint')
            sys.exit(-1)
        resume_folder = opt.resume if os.path.isdir(opt.resume) else os.path.dirname(opt.resume)
        opt.config = os.path.normpath(os.path.join(resume_folder, '../config.yaml'))
        if os.path.exists(opt.config):
            logger.info(f'Using config file from checkpoint directory: {opt.config}')
        else:
            logger.error('Config not found in resume directory')
            sys.exit(-2)
    elif rank == 0:
        with open(opt.config, 'r') as in_fid, open(os.path.join(opt.log_path, 'config.yaml'), 'w') as out_fid:
            out_fid.write(f'# Original file name: {opt.config}\n')
            out_fid.write(in_fid.read())

    from regtr_amd import RegTR, load_config
    from regtr_amd import harness
    # torch sizes its intra-op thread team from os.cpu_count(); inside a CPU-quota container that is far more threads than cores it may
    # use, and the oversubscribed team gets the whole process throttled (the host work here is launch issue, not CPU tensor math)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), harness.usable_cores() // 2, 8)))
    cfg = load_config(opt.config)
    if opt.preprocessor:
        cfg.update({'kpconv_neighbor_order': 'index' if opt.preprocessor == 'gpu' else 'nearest',
                    'kpconv_voxel_key': 'floor' if opt.preprocessor == 'gpu' else 'origin'})
    if opt.neighbor_order:
        cfg.update({'kpconv_neighbor_order': opt.neighbor_order})
    if opt.voxel_key:
        cfg.update({'kpconv_voxel_key': opt.voxel_key})
    if cfg.dataset == '3dmatch':
        assert opt.benchmark in ['3DMatch', '3DLoMatch'], "Benchmark for 3dmatch dataset must be one of ['3DMatch', '3DLoMatch']"
        cfg.benchmark = opt.benchmark
    elif cfg.dataset == 'modelnet':
        assert opt.benchmark in ['ModelNet', 'ModelLoNet'], "Benchmark for modelnet dataset must be one of ['ModelNet', 'ModelLoNet']"
        cfg.partial = [0.7, 0.7] if opt.benchmark == 'ModelNet' else [0.5, 0.5]

    if not torch.cuda.is_available():
        logger.error('regtr_amd runs on an MI355X (HIP) device only; there is no CPU path')
        sys.exit(-3)
    device = torch.device('cuda', local_rank)
    # loader processes: forked HERE -- before this process holds a HIP context (torch.cuda.is_available() only counts devices), before the
    # optional pool reservation, before torch.distributed and before the model exists, so the workers inherit none of that.  They sit
    # idle until run_test hands them batches (nothing is read ahead of the timed loop); the pair source reaches them by file later.
    workers = opt.num_workers if opt.num_workers >= 0 else 4
    pool = harness.LoaderPool(None, device, workers=workers, max_batch=opt.batch) if workers > 0 else None
    if pool is not None:
        import atexit
        atexit.register(pool.close)      # every sys.exit() / exception path below: workers ended, the pair-source file removed (close is idempotent)
    torch.cuda.set_device(device)
    if opt.alloc_conf:
        torch.cuda.memory._set_allocator_settings(opt.alloc_conf)
    reserve_gb = max(opt.reserve_gb, 0.0)
    if reserve_gb > 0:
        # optional: one large block reserved up front and handed back to torch's caching allocator, so that later requests are carved out
        # of it instead of reaching hipMalloc (every level size of every batch is data dependent)
        torch.empty(int(reserve_gb * 2**30), dtype=torch.uint8, device=device)
        logger.info(f'allocator pool pre-sized with a {reserve_gb:.0f} GiB block')
    if world > 1 or 'RANK' in os.environ:      # under torch.distributed.run: a process group at every world size (one code path)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=device)

    # pairs
    if opt.synthetic > 0 and opt.materialize:
        info = harness.materialize_synthetic(opt.materialize, opt.synthetic, overlap=opt.overlap, logger=logger if rank == 0 else None,
                                             rank=rank, world=world, distinct=opt.distinct)
        pairs = harness.ThreeDMatchPairs(info, opt.materialize, cache_dir=opt.cache_dir)
    elif opt.synthetic > 0:
        pairs = harness.SyntheticPairs(opt.synthetic, points=20000 if cfg.dataset == '3dmatch' else 717, overlap=opt.overlap)
    elif cfg.dataset == '3dmatch':
        info = opt.info or os.path.join('datasets', '3dmatch', f'test_{opt.benchmark}_info.pkl')
        cfg_root = cfg.get('root', None)         # the shipped regtr_amd/conf/*.yaml carry no dataset root: --data_root supplies it
        roots = [opt.data_root] if opt.data_root else ([cfg_root] if isinstance(cfg_root, str) else list(cfg_root or []))
        root = next((r for r in roots if os.path.exists(os.path.join(r, 'test'))), None)
        if root is None or not os.path.exists(info):
            logger.error(f'Dataset not found (info {info}, roots {roots}); pass --data_root / --info or use --synthetic N')
            sys.exit(-4)
        pairs = harness.ThreeDMatchPairs(info, root, cache_dir=opt.cache_dir)
    else:
        logger.error('ModelNet h5 loading is not part of the inference hot path here (h5py is not a dependency); use --synthetic N')
        sys.exit(-4)

    if pool is not None:
        pool.set_source(pairs)           # (a pickle file the workers load on their first task)
        pool.prepare()                   # page-locked staging buffers, outside the timed loop

    model = RegTR(cfg).to(device)
    if opt.resume:
        state = torch.load(opt.resume, map_location=device, weights_only=False)
        missing = model.load_state_dict(state['state_dict'], strict=False)       # CheckPointManager.load (torch_helpers.py:222)
        logger.info(f'Loaded checkpoint {opt.resume} (step {state.get("step", "?")}); {missing}')
    else:
        logger.warning('No checkpoint given. Will perform inference using random weights')
    # forwards in flight (round 6): replicas with the same weights, one per host thread / HIP stream (harness.run_test)
    from regtr_amd.workload import replicate
    n_rep = opt.replicas if opt.replicas > 0 else 3      # (also for --batch 1, the reference's loop: 349 -> 607 pairs/s end to end, profiles/r06_y_*)
    models = replicate(model.eval(), cfg, n_rep, device)

    if not opt.no_warmup:
        # one untimed forward on a synthetic batch of the run's shape: the weights' one-time re-layout (split planes), the first growth
        # of the caching allocator (~8 GB for 64 pairs) and the tile plans -- what bench.py's warm-up steps absorb.  No dataset file is
        # touched before the timed loop.
        from regtr_amd.synthetic import synth_modelnet_pair, synth_pair
        t_w = time.perf_counter()
        gen = [(synth_pair(900001 + i, opt.warmup_points) if cfg.dataset == '3dmatch' else synth_modelnet_pair(900001 + i)) for i in range(2)]
        wb = {'src_xyz': [torch.from_numpy(gen[i % 2][0]).to(device) for i in range(opt.batch)],
              'tgt_xyz': [torch.from_numpy(gen[i % 2][1]).to(device) for i in range(opt.batch)]}
        for m in models:             # every replica: its weights' re-layout, its workspaces
            m.eval()
            with torch.no_grad():
                m(wb)
        torch.cuda.synchronize(device)
        del wb
        logger.info(f'warm-up forward ({opt.batch} synthetic pairs, untimed): {time.perf_counter() - t_w:.2f} s')
    t_run = time.perf_counter()
    poses, ids, timing = harness.run_test(models if len(models) > 1 else model, pairs, opt.batch, device, logger, opt.max_pairs, loader_pool=pool)
    if pool is not None:
        logger.info(f'loader: {timing["loader"]}; forward ms (host clock, incl. the end-of-forward status wait): {timing["forward_ms"]}')
        pool.close()
    from_files = isinstance(pairs, harness.ThreeDMatchPairs)
    if rank == 0:
        recs, gts = [], []
        for pose, i in zip(poses, ids):
            meta = pairs[int(i)] if not from_files else {'src_path': pairs.infos['src'][int(i)], 'tgt_path': pairs.infos['tgt'][int(i)],
                                                            'pose': np.concatenate([pairs.infos['rot'][int(i)], pairs.infos['trans'][int(i)].reshape(3, 1)], 1)}
            recs.append({'src_path': meta['src_path'], 'tgt_path': meta['tgt_path'], 'pose': pose})
            gts.append(meta['pose'])
        if cfg.dataset == '3dmatch':
            harness.write_est_log(opt.log_path, opt.benchmark, recs)
            t_all = time.perf_counter() - t_run
            logger.info(f'est.log files written under {os.path.join(opt.log_path, opt.benchmark)}')
            logger.info(f'[End to end] {len(ids)} pairs, {"files -> " if from_files else "generator -> "}H2D -> forward -> pose gather -> est.log: '
                        f'{t_all:.2f} s = {len(ids) / t_all:.1f} pairs/s on {timing["world"]} GPU(s), batch {opt.batch}, {len(models)} forward(s) in flight, {workers} loader process(es)'
                        f'{", .npy cache" if opt.cache_dir else ""}')
            gt_folder = os.path.join(opt.benchmark_dir, opt.benchmark)
            complete = opt.max_pairs is None or opt.max_pairs <= 0 or opt.max_pairs >= len(pairs)
            if opt.synthetic == 0 and os.path.isdir(gt_folder) and not complete:
                logger.warning('partial run (--max_pairs): the registration recall needs every pair of the benchmark; skipping the evaluation')
            if opt.synthetic == 0 and os.path.isdir(gt_folder) and complete:
                # Evaluate 3DMatch registration recall (generic_reg_model.py:180-186), in process
                from regtr_amd.evaluation import benchmark
                results_str, mean_recall = benchmark(os.path.join(opt.log_path, opt.benchmark), gt_folder)
                logger.info('\n' + results_str)
                logger.info(f'Mean registration recall: {mean_recall:.4f}')
        else:
            np.save(os.path.join(opt.log_path, 'pred_transforms.npy'), poses[:, None])    # (n, 1, 3, 4) like torch.stack of (B=1,3,4)
            # RPMNet / DCP metrics (benchmark_modelnet.py:33-105); the clean cloud of a synthetic pair is its own target
            from regtr_amd.evaluation import modelnet_metrics, summarize_metrics
            items = [pairs[int(i)] for i in ids]
            per = [modelnet_metrics(poses[k:k + 1], np.stack(gts)[k:k + 1], it['src_xyz'][None], it['tgt_xyz'][None], it['tgt_xyz'][None])
                   for k, it in enumerate(items)]
            sm = summarize_metrics({key: np.concatenate([p[key] for p in per]) for key in per[0]})
            logger.info('DeepCP metrics:{:.4f}(rot-rmse) | {:.4f}(rot-mae) | {:.4g}(trans-rmse) | {:.4g}(trans-mae)'.format(
                sm['r_rmse'], sm['r_mae'], sm['t_rmse'], sm['t_mae']))
            logger.info('Rotation error {:.4f}(deg, mean) | {:.4f}(deg, rmse)'.format(sm['err_r_deg_mean'], sm['err_r_deg_rmse']))
            logger.info('Translation error {:.4g}(mean) | {:.4g}(rmse)'.format(sm['err_t_mean'], sm['err_t_rmse']))
            logger.info('Chamfer error: {:.7f}(mean-sq)'.format(sm['chamfer_dist']))
        rot, trans = harness.pose_errors(poses, np.stack(gts))
        ok = np.logical_and(rot < cfg.get('reg_success_thresh_rot', 10), trans < cfg.get('reg_success_thresh_trans', 0.1))
        logger.info(f'[Metrics] rot_err_deg_final: {rot.mean():.4f}, trans_err_final: {trans.mean():.4f}, reg_success_final: {ok.mean():.4f} '
                    f'({len(ids)} pairs, {timing["pairs"] / timing["elapsed_s"]:.1f} pairs/s on {timing["world"]} GPU(s))')
    if world > 1 or 'RANK' in os.environ:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
