"""Measurement behind bench.py's `roofline*`, `forward_*` and `preprocess` blocks (moved out of bench.py in round 6): per-launch HIP-event timing of the
KPConv gathers / dense products / attention core inside real forwards run on ONE stream, the pyramid alone, the code-version stamp, and the
rocprofv3 counter passes (`python bench.py --collect-pmc`) with their per-launch and per-forward HBM byte tables.  Nothing here imports oracle/."""
import json
import os
import subprocess
import sys

import numpy as np
import torch

from .workload import DEFAULT_PAIRS, HBM_PEAK_GBS, MFMA_BF16_PEAK_TFS, kpconv_algorithmic_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')      # the counter passes run rocprofv3 over this script


class one_stream:
    """`with one_stream():` -- the forwards inside run on ONE stream (RegTR's second, pyramid stream off).  The per-launch event timings of the
    roofline blocks are taken this way since round 5: with the pyramid next to them a level-0 launch shared the chip with the radius kernels while
    its events ran (the in-run gather rate then sat 7-8 % under the rocprofv3 mean of the same kernels, which serialises the streams)."""

    def __enter__(self):
        from regtr_amd import regtr as regtr_mod
        self.mod, self.prev = regtr_mod, regtr_mod.overlap_preprocessing
        regtr_mod.overlap_preprocessing = False

    def __exit__(self, *exc):
        self.mod.overlap_preprocessing = self.prev
        return False


def measure_kpconv_roofline(model, batch, reps=5, keep=None):
    """Times every KPConv gather launch (k_kpconv_gather_*) with HIP events on the stream it is enqueued on (torch's current stream) during
    real forwards run on one stream; achieved = sum of algorithmic bytes / sum of durations."""
    from regtr_amd import context
    records = []
    with one_stream(), context.recording(gather_records=records):
        for _ in range(reps):
            model({'src_xyz': list(batch['src_xyz']), 'tgt_xyz': list(batch['tgt_xyz'])})
        torch.cuda.synchronize()
    if keep is not None:
        keep['gather'] = (records, reps)
    t_gather = sum(r[0].elapsed_time(r[1]) for r in records) * 1e-3
    t_gemm = sum(r[1].elapsed_time(r[2]) for r in records) * 1e-3
    alg = sum(kpconv_algorithmic_bytes(r[3], r[4], r[5], r[6]) for r in records)
    # the gather kernel's share of B_kp: index + neighbour xyz + neighbour feature rows + query xyz (reads only; the
    # WF intermediate it writes is an implementation artefact, not algorithmic traffic)
    alg_gather = sum(r[3] * r[4] * (4 + 12 + 4 * r[5]) + r[3] * 12 for r in records)
    n_launch = len(records)
    shapes = {}
    for r in records:
        d = shapes.setdefault((r[3], r[4], r[5]), [0.0, 0])
        d[0] += r[0].elapsed_time(r[1]) * 1e-3; d[1] += 1
    by_shape = [{'queries': nq, 'H': H, 'Cin': cin, 'launches_per_step': n // reps, 'us': round(t / n * 1e6, 1),
                 'alg_GBs': round((nq * H * (4 + 12 + 4 * cin) + nq * 12) / (t / n) / 1e9), 'kernel': 'k_kpconv_gather_c1p' if cin == 1 else 'k_kpconv_gather_mfma'}
                for (nq, H, cin), (t, n) in sorted(shapes.items(), key=lambda kv: -kv[1][0])]
    return {
        'kernel': 'k_kpconv_gather_* (every KPConv gather launch of a forward, the first block\'s Cin = 1 gather included)',
        'launches_per_step': n_launch // reps, 'by_shape': by_shape,
        'avg_launch_us': t_gather / n_launch * 1e6,
        'achieved_gather_kernel_GBs': alg_gather / t_gather / 1e9,
        'achieved_kpconv_op_GBs': alg / (t_gather + t_gemm) / 1e9,
        'alg_bytes_per_step': alg / reps, 'alg_gather_bytes_per_step': alg_gather / reps,
        'gather_s_per_step': t_gather / reps, 'gemm_s_per_step': t_gemm / reps,
        'timing': 'HIP events on the launch stream around every gather launch of real forwards run on ONE stream (nothing else on the chip while a launch is timed)',
    }


def measure_preprocess(model, batch, reps=5):
    """The preprocessing pyramid ALONE (grid subsampling + radius neighbours of every level: kpconv.py:426-537), event-timed on one stream
    -- in a forward most of it runs under the level-0 convolutions on the second stream, so this is its cost, not its exposed time."""
    clouds = list(batch['src_xyz']) + list(batch['tgt_xyz'])
    model.preprocessor(clouds)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        meta = model.preprocessor(clouds)
    e1.record()
    torch.cuda.synchronize()
    return {'pyramid_ms_alone': round(e0.elapsed_time(e1) / reps, 3), 'level_points': [int(p.shape[0]) for p in meta['points']],
            'what': 'grid subsampling + conv / pool radius-neighbour tables of every level, one stream, nothing overlapped'}


def measure_attention(model, batch, n_heads, d_embed, n_layers, reps=5):
    """Times every attention-core launch (k_mha_fwd*) with HIP events on its stream during real forwards.  Algorithmic flops
    (SURVEY.md 8d): per layer and pair 4 d (Ns^2 + Nt^2 + 2 Ns Nt) -- QK^T and AV of the two self- and the two
    cross-attentions, d = d_embed."""
    from regtr_amd import context
    records = []
    with one_stream(), context.recording(mha_records=records):
        for _ in range(reps):
            b = {'src_xyz': list(batch['src_xyz']), 'tgt_xyz': list(batch['tgt_xyz'])}
            model(b)
        torch.cuda.synchronize()
    lens = b['kpconv_meta']['_lens_host'][-1]
    B = len(lens) // 2
    flops_fwd = n_layers * sum(4.0 * d_embed * (lens[i] ** 2 + lens[B + i] ** 2 + 2.0 * lens[i] * lens[B + i]) for i in range(B))
    t = sum(e0.elapsed_time(e1) for e0, e1 in records) * 1e-3
    return {'kernel': 'k_mha_fwd', 'launches_per_step': len(records) // reps, 'avg_launch_us': t / len(records) * 1e6,
            'alg_flops_per_step': flops_fwd, 'attention_s_per_step': t / reps, 'achieved_TFs': flops_fwd * reps / t / 1e12,
            'tokens_per_cloud_mean': float(np.mean(lens))}


def measure_gemm_roofline(model, batch, reps=3, keep=None):
    """Times every dense launch (split GEMM in either format, one-shot strip, block tail, exact-f32) with HIP events on its stream
    during real forwards and prices each against BOTH rooflines: matrix pipe = 2 M N K x terms issued / 2.5 PFLOP/s (the f16 pair split
    issues 3 MFMA terms per product, bf16x3 six; the exact-f32 MFMA runs at 157.3 TFLOP/s) and HBM = (4 M K [x passes] + 4 M N +
    weight bytes) / 8 TB/s; a launch's bound is the larger of the two times.  -> the `roofline_gemm` block of the bench line."""
    from regtr_amd import context
    records = []
    # ONE stream while the launches are timed (round 5): with the pyramid on the second stream the level-0 products shared the chip with the
    # radius kernels while their events ran, and their fractions were pessimistic by an unknown amount
    with one_stream(), context.recording(gemm_records=records):
        for _ in range(reps):
            model({'src_xyz': list(batch['src_xyz']), 'tgt_xyz': list(batch['tgt_xyz'])})
        torch.cuda.synchronize()
    if keep is not None:
        keep['gemm'] = (records, reps)
    shapes = {}
    for e0, e1, m in records:
        key = (m['route'], m['M'], m['N'], m['K'], m['fold'], m['stats'])
        d = shapes.setdefault(key, {'t': 0.0, 'n': 0, 'meta': m})
        d['t'] += e0.elapsed_time(e1) * 1e-3
        d['n'] += 1
    rows, tot, tot_bound, by_route = [], 0.0, 0.0, {}
    for (route, M, N, K, fold, stats), d in shapes.items():
        m = d['meta']
        t = d['t'] / d['n']
        flops = 2.0 * M * N * K
        t_mfma = (flops * m['terms'] / (MFMA_BF16_PEAK_TFS * 1e12)) if m['terms'] else flops / 157.3e12
        bytes_ = 4.0 * M * K * m.get('passes', 1) + 4.0 * M * N + m['w_bytes'] * K * N
        t_hbm = bytes_ / (HBM_PEAK_GBS * 1e9)
        bound = 'mfma' if t_mfma >= t_hbm else 'hbm'
        per_step = d['n'] / reps
        tot += t * per_step; tot_bound += max(t_mfma, t_hbm) * per_step
        r = by_route.setdefault(route, [0.0, 0])
        r[0] += t * per_step; r[1] += per_step
        rows.append({'route': route, 'M': M, 'N': N, 'K': K, 'folded_norm_operand': bool(fold), 'stats_epilogue': bool(stats),
                     'launches_per_step': per_step, 'us': round(t * 1e6, 1), 'bound': bound, 'frac': round(max(t_mfma, t_hbm) / t, 3),
                     'TFLOPs_f32_equiv': round(flops / t / 1e12, 1), 'GBs': round(bytes_ / t / 1e9)})
    rows.sort(key=lambda r: -r['us'] * r['launches_per_step'])
    return {'what': 'every dense contraction of a forward (KPConv kernel-point contractions, unary / shortcut / projection / FFN / head Linears), '
                    'event-timed per launch with the forward on ONE stream (nothing else on the chip while a launch is timed); frac = roofline time (the larger of matrix-pipe and HBM time) / measured time',
            'ms_per_step': round(tot * 1e3, 3), 'roofline_ms_per_step': round(tot_bound * 1e3, 3), 'frac': round(tot_bound / tot, 3),
            'launches_per_step': sum(r['launches_per_step'] for r in rows),
            'by_route_ms': {k: round(v[0] * 1e3, 3) for k, v in sorted(by_route.items(), key=lambda kv: -kv[1][0])},
            'peaks': {'mfma_16bit_dense_TFLOPs': MFMA_BF16_PEAK_TFS, 'mfma_f32_TFLOPs': 157.3, 'hbm_GBs': HBM_PEAK_GBS},
            'top_shapes': rows[:14]}


def code_version():
    """{source_sha256, git_commit, lib_sha256}: the kernel sources this tree holds (regtr_amd/build.py: source_hash), the commit the
    library was built at (regtr_amd/_build_info.json, written by the build where .git exists) and the loaded library's own hash."""
    import hashlib
    from regtr_amd import _lib
    from regtr_amd.build import source_hash
    v = {'source_sha256': source_hash(), 'git_commit': None, 'git_dirty': None, 'lib_sha256': None}
    try:
        info = json.load(open(os.path.join(ROOT, 'regtr_amd', '_build_info.json')))
        if info.get('source_sha256') == v['source_sha256']:
            v['git_commit'], v['git_dirty'] = info.get('git_commit'), info.get('git_dirty')
    except (OSError, ValueError):
        pass
    try:
        v['lib_sha256'] = hashlib.sha256(open(_lib.LIB_PATH, 'rb').read()).hexdigest()
    except OSError:
        pass
    return v


def pmc_traffic(pairs, points, shuffle, detail, real=False):
    """HBM bytes per KPConv-gather launch from profiles/pmc_traffic.json -- written by `python bench.py --collect-pmc` (counters cannot
    be read from inside the timed process: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over this very script).  Reported only
    when that file was taken on THIS workload and THIS code version (kernel source hash); otherwise null, with the reason."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        detail['traffic_note'] = 'profiles/pmc_traffic.json absent: run `python bench.py --collect-pmc` on the GPU'
        return None
    if real or t.get('workload') != {'pairs': pairs, 'points': points, 'shuffle': bool(shuffle)}:
        detail['traffic_note'] = f"profiles/pmc_traffic.json was taken on another workload ({t.get('workload')})"
        return None
    here = code_version()
    if t.get('code', {}).get('source_sha256') != here['source_sha256']:
        detail['traffic_note'] = (f"profiles/pmc_traffic.json was taken on other kernel sources (commit {t.get('code', {}).get('git_commit')}): "
                                  're-run `python bench.py --collect-pmc`')
        return None
    detail['traffic_unit'] = 'HBM bytes per launch (mean over the gather launches of a forward)'
    detail['traffic_source'] = t.get('source')
    detail['traffic_code'] = t.get('code')
    detail['alg_bytes_per_launch'] = detail['alg_gather_bytes_per_step'] / detail['launches_per_step']
    # the candid companion of `frac`: bytes the counters saw MOVE through HBM per launch / launch time / peak.  The 40x re-read feature rows are
    # served by L2 / Infinity Cache, so this is far below the algorithmic fraction -- the gather is not an HBM stream (DESIGN sections 3, 8)
    detail['counter_hbm_GBs'] = t['hbm_bytes_per_launch'] / (detail['avg_launch_us'] * 1e-6) / 1e9
    detail['counter_hbm_frac_of_peak'] = detail['counter_hbm_GBs'] / HBM_PEAK_GBS
    detail['traffic_over_algorithmic'] = t['hbm_bytes_per_launch'] / detail['alg_bytes_per_launch']
    return t['hbm_bytes_per_launch']


def forward_traffic(pairs, points, shuffle, real, compulsory_bytes):
    """`forward_traffic` of the bench line: HBM bytes ONE forward moves (every kernel, from the stamped counter file of `--collect-pmc`) against
    its compulsory bytes (regtr_amd/workload.py: forward_compulsory_bytes -- SURVEY.md Appendix B on this batch's level sizes) -- so that the
    ratio (round 5: 260.7 GB against 23 GB, 12 x: the WF intermediate, the InstanceNorm passes, the split GEMMs' operands) is visible in every
    line and cannot regress silently.  hbm_GB is null (with the reason) when the counter file is not of this workload and these kernel sources."""
    out = {'hbm_GB': None, 'compulsory_GB': round(compulsory_bytes / 1e9, 3), 'ratio': None, 'top3': None,
           'compulsory': 'every array of the path touched once: SURVEY.md Appendix B per KPConv block + preprocessing + tokens + weights, on this batch\'s level sizes'}
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
    except (OSError, ValueError):
        out['note'] = 'profiles/pmc_traffic.json absent: run `python bench.py --collect-pmc` on the GPU'
        return out
    fw = t.get('forward')
    if real or t.get('workload') != {'pairs': pairs, 'points': points, 'shuffle': bool(shuffle)} or not fw:
        out['note'] = f"profiles/pmc_traffic.json was taken on another workload ({t.get('workload')})" if fw else 'profiles/pmc_traffic.json predates the per-forward totals'
        return out
    if t.get('code', {}).get('source_sha256') != code_version()['source_sha256']:
        out['note'] = f"profiles/pmc_traffic.json was taken on other kernel sources (commit {t.get('code', {}).get('git_commit')}): re-run `python bench.py --collect-pmc`"
        return out
    out.update(hbm_GB=round(fw['hbm_bytes_per_forward'] / 1e9, 2), fetch_GB=round(fw['fetch_bytes_per_forward'] / 1e9, 2),
               write_GB=round(fw['write_bytes_per_forward'] / 1e9, 2), ratio=round(fw['hbm_bytes_per_forward'] / compulsory_bytes, 2),
               top3=fw['top3'], forwards_counted=fw['forwards'], source=t.get('source'), code=t.get('code'))
    return out


def collect_pmc(args):
    """`python bench.py --collect-pmc`: the counter passes behind `roofline.traffic`, reproducibly.  Two rocprofv3 runs (FETCH_SIZE, then
    WRITE_SIZE: separate passes, kernel trace only -- MI355X_MICROARCH.md's HBM recipe) over `bench.py --steps 2 --warmup 1` on the same
    workload flags; per-kernel HBM bytes per launch = 2 x FETCH_SIZE KB (gfx950 tallies 128-byte requests at 64 B) + WRITE_SIZE KB.
    Writes profiles/pmc_traffic.json (stamped with the code version) and profiles/<tag>_pmc_traffic_kernels.md."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    from collections import defaultdict
    if not shutil.which('rocprofv3'):
        sys.exit('bench.py --collect-pmc: rocprofv3 not found')
    work = tempfile.mkdtemp(prefix='regtr_pmc_', dir=os.environ.get('TMPDIR', '/tmp'))
    inner = [sys.executable, BENCH, '--steps', '2', '--warmup', '1', '--settle-s', '0', '--no-cpu-baseline', '--no-roofline', '--no-real', '--replicas', '1', '--head-init', 'uniform',      # (uniform head: no calibration forward among the counted ones)
            
             '--parity-pairs', '0', '--no-strict-f32', '--points', str(args.points)] + (['--pairs', str(args.pairs)] if args.pairs else []) \
        + (['--shuffle'] if args.shuffle else [])
    vals = defaultdict(lambda: defaultdict(list))
    for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
        out = os.path.join(work, ctr)
        cmd = ['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', out, '-o', 'p', '--'] + inner
        r = subprocess.run(cmd, cwd=work, env=dict(os.environ, TMPDIR=work), capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(f'bench.py --collect-pmc: {ctr} pass failed:\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}')
        for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
            per, names = defaultdict(float), {}
            for row in csv.DictReader(open(f)):
                if row['Counter_Name'] == ctr:
                    per[row['Dispatch_Id']] += float(row['Counter_Value'])
                    name = re.sub(r'\(anonymous namespace\)::', '', row['Kernel_Name'])
                    m = re.match(r'(?:void )?([\w:<>, ]+?)\(', name)
                    names[row['Dispatch_Id']] = (m.group(1) if m else name)[:80]
            for d, v in per.items():
                vals[names[d]][ctr].append(v)
    shutil.rmtree(work, ignore_errors=True)
    kernels = {}
    for k, v in vals.items():
        f, w = v.get('FETCH_SIZE', []), v.get('WRITE_SIZE', [])
        kernels[k] = {'launches': max(len(f), len(w)), 'fetch_bytes_per_launch': 2 * 1024 * sum(f) / max(len(f), 1),
                      'write_bytes_per_launch': 1024 * sum(w) / max(len(w), 1)}
    # per-forward totals over EVERY kernel: forwards = launches of the once-per-forward pose kernel
    n_fwd = max(kernels.get('k_procrustes', {}).get('launches', 0), 1)
    per_fwd = {k: v['launches'] * (v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']) / n_fwd for k, v in kernels.items()}
    forward = {'forwards': n_fwd,
               'fetch_bytes_per_forward': sum(v['launches'] * v['fetch_bytes_per_launch'] for v in kernels.values()) / n_fwd,
               'write_bytes_per_forward': sum(v['launches'] * v['write_bytes_per_launch'] for v in kernels.values()) / n_fwd,
               'hbm_bytes_per_forward': sum(per_fwd.values()),
               'top3': [{'kernel': k, 'GB': round(b / 1e9, 2), 'launches_per_forward': round(kernels[k]['launches'] / n_fwd, 1)}
                        for k, b in sorted(per_fwd.items(), key=lambda kv: -kv[1])[:3]]}
    g = {k: v for k, v in kernels.items() if 'k_kpconv_gather' in k}
    n = sum(v['launches'] for v in g.values())
    if n == 0:
        sys.exit('bench.py --collect-pmc: no KPConv gather launch in the counter output')
    total = sum(v['launches'] * (v['fetch_bytes_per_launch'] + v['write_bytes_per_launch']) for v in g.values())
    code = code_version()
    pairs = args.pairs if args.pairs else DEFAULT_PAIRS['3dmatch']
    res = {'workload': {'pairs': pairs, 'points': args.points, 'shuffle': bool(args.shuffle)}, 'hbm_bytes_per_launch': total / n, 'code': code,
           'source': ('`python bench.py --collect-pmc`: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only) over '
                      f'`bench.py --steps 2 --warmup 1`; bytes = 2 x FETCH_SIZE KB (gfx950 tallies 128-B requests at 64 B) + WRITE_SIZE KB; mean '
                      f'over {n} gather launches; kernel sources {code["source_sha256"][:12]}, commit {code["git_commit"]}'),
           'gather_kernels': g, 'forward': forward}
    os.makedirs(os.path.join(ROOT, 'profiles'), exist_ok=True)
    with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w') as f:
        json.dump(res, f, indent=1)
    with open(os.path.join(ROOT, 'profiles', f'{args.pmc_tag}_pmc_traffic_kernels.md'), 'w') as f:
        f.write(f'# HBM traffic per launch (rocprofv3 PMC, `python bench.py --collect-pmc`), kernel sources {code["source_sha256"][:12]}, commit {code["git_commit"]}\n\n')
        f.write('| kernel | launches | fetch MB / launch | write MB / launch |\n|---|---|---|---|\n')
        for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['launches'] * (kv[1]['fetch_bytes_per_launch'] + kv[1]['write_bytes_per_launch'])):
            f.write(f"| {k} | {v['launches']} | {v['fetch_bytes_per_launch'] / 1e6:.1f} | {v['write_bytes_per_launch'] / 1e6:.1f} |\n")
    print(json.dumps({k: res[k] for k in ('workload', 'hbm_bytes_per_launch', 'forward', 'code')}))


