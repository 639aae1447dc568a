"""Micro-benchmark of regtr_kpconv_gather on the real neighbour tables of the bench workload (run on the MI355X box):
    [REGTR_VARIANT=name] python tools/gather_bench.py [--pairs 16]
Prints per level: launch time, algorithmic GB/s (Nq*H*(4+12+4*Cin) + 12*Nq bytes per launch)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from regtr_amd import _lib, load_config, ops  # noqa: E402
from regtr_amd.kpconv import Preprocessor  # noqa: E402
from regtr_amd.kernel_points import load_kernels  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--pairs', type=int, default=16)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--no-fold', action='store_true', help='features already normalised: no InstanceNorm+LeakyReLU fold in the gather')
    ap.add_argument('--pre', action='store_true', help='features final + precomputed row flags (the encoder\'s form): no fold, no row sums')
    ap.add_argument('--debug', action='store_true')
    ap.add_argument('--xpat', default='', help='debug feature pattern: ones | chan | row')
    ap.add_argument('--levels', default='', help='comma list of level indices 0..6 to run')
    ap.add_argument('--against', default='', help='second build (libregtr_hip.<name>.so) to run on the same inputs: max relative WF difference')
    args = ap.parse_args()
    if args.pre:
        args.no_fold = True
    dev = torch.device('cuda', 0)
    cfg = load_config(os.path.join(bench.ROOT, 'regtr_amd', 'conf', '3dmatch.yaml'))
    pairs = [bench.synth_pair(i, 20000) for i in range(args.pairs)]
    pts = [torch.from_numpy(s).to(dev) for s, _ in pairs] + [torch.from_numpy(t).to(dev) for _, t in pairs]
    meta = Preprocessor(cfg)(pts)
    L = _lib.lib()
    L2 = None
    if args.against:
        import ctypes
        L2 = ctypes.PyDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), f'libregtr_hip.{args.against}.so'))
        L2.regtr_kpconv_gather.restype, L2.regtr_kpconv_gather.argtypes = _lib.SIGNATURES['regtr_kpconv_gather']
    torch.manual_seed(0)
    r0 = cfg.first_subsampling_dl * cfg.conv_radius
    total = 0.0
    for lvl, (Cin, strided) in enumerate([(32, False), (32, True), (64, False), (64, True), (128, False), (128, True), (256, False), (1, False)]):      # (level 7: the first block, Cin = 1)
        layer = [0, 0, 1, 1, 2, 2, 3, 0][lvl]
        if args.levels and str(lvl) not in args.levels.split(','):
            continue
        s_pts = meta['points'][layer]
        q_pts = meta['points'][layer + 1] if strided else s_pts
        nbr = meta['_pools_i32'][layer] if strided else meta['_neighbors_i32'][layer]
        seg_s, seg_q = meta['_seg_off'][layer], meta['_seg_off'][layer + 1 if strided else layer]
        ns, nq, H = s_pts.shape[0], q_pts.shape[0], nbr.shape[1]
        radius = r0 * 2 ** layer
        kp = torch.tensor(load_kernels(radius, 15, dimension=3, fixed='center'), dtype=torch.float32, device=dev)
        x = torch.randn(ns, Cin, device=dev) if Cin > 1 else torch.ones(ns, 1, device=dev)
        if args.xpat == 'ones':
            x = torch.ones(ns, Cin, device=dev)
        elif args.xpat == 'chan':
            x = (torch.arange(Cin, device=dev, dtype=torch.float32) + 1).repeat(ns, 1).contiguous()
        elif args.xpat == 'row':
            x = (torch.arange(ns, device=dev, dtype=torch.float32) % 64 + 1)[:, None].repeat(1, Cin).contiguous()
        st = ops.instnorm_stats(x, seg_s, max(meta['_lens_host'][layer])) if Cin > 1 else None
        wf = torch.empty(nq, 15 * Cin, device=dev); num = torch.empty(nq, device=dev)
        flag = torch.cat((s_pts, (x.sum(1, keepdim=True) > 0).float()), 1).contiguous() if args.pre else None

        def run(L=L, wf=wf, num=num):
            _lib.check(L.regtr_kpconv_gather(_lib.ptr(q_pts), nq, _lib.ptr(s_pts), ns, _lib.iptr(nbr), H, _lib.ptr(x), Cin, None, _lib.ptr(flag),
                                             _lib.ptr(kp), 15, radius * 0.8, None if args.no_fold else _lib.ptr(st), None if args.no_fold else _lib.iptr(seg_q), 0 if args.no_fold else seg_q.numel() - 1, 0.1,
                                             _lib.ptr(wf), 0, _lib.ptr(num), _lib.stream()), 'gather')
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / args.reps * 1e3
        alg = nq * H * (16 + 4 * Cin) + 12 * nq
        total += us
        print(f'layer {layer} {"pool" if strided else "conv"} Cin={Cin:3d} nq={nq:7d} ns={ns:7d}: {us:8.1f} us  {alg / us / 1e3:7.0f} GB/s alg '
              f'(+{nq * 15 * Cin * 4 / us / 1e3:5.0f} GB/s WF write)  chk={float(wf.sum()):.6e} {float(num.sum()):.1f}')
        if L2 is not None:
            wf2 = torch.empty_like(wf); num2 = torch.empty_like(num)
            run(L2, wf2, num2)
            torch.cuda.synchronize()
            scale = float(wf2.abs().max())
            print(f'   vs {args.against}: max abs diff {float((wf - wf2).abs().max()):.3e} (max |wf| {scale:.3e}), num equal {bool((num == num2).all())}')
            if args.debug and float((wf - wf2).abs().max()) > 1e-3:
                bad = ((wf - wf2).abs() > 1e-3).view(nq, 15, Cin)
                rows = bad.any(2).any(1)
                print('   bad queries', int(rows.sum()), 'of', nq, 'first', rows.nonzero()[:12, 0].tolist())
                print('   bad by q % 32:', bad.any(2).any(1).view(-1)[:nq // 32 * 32].view(-1, 32).sum(0).tolist())
                print('   bad by kernel point:', bad.any(2).sum(0).tolist())
                print('   bad by channel:', bad.any(1).sum(0).tolist())
                q = int(rows.nonzero()[0, 0])
                print('   query', q, 'nbr', nbr[q].tolist())
                print('   got ', wf.view(nq, 15, Cin)[q, 0, :8].tolist(), '\n   want', wf2.view(nq, 15, Cin)[q, 0, :8].tolist())
                print('   got ', wf.view(nq, 15, Cin)[q, 3, 40:52].tolist(), '\n   want', wf2.view(nq, 15, Cin)[q, 3, 40:52].tolist())
            del wf2, num2
    print(f'total {total:.1f} us  variant={os.environ.get("REGTR_VARIANT", "")}')


if __name__ == '__main__':
    main()
