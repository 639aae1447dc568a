"""GPU: the split GEMM on RegTR's shapes at 64 pairs per forward -- tiled kernel (REGTR_X3_STRIP=0) against the row-strip kernel
(default), each in its own process (the switch is read once), same random operands, error against a float64 product.
    python tools/x3_bench.py            (both arms)      python tools/x3_bench.py --arm   (this process only)"""
import json
import os
import subprocess
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SHAPES = [  # (M, N, K, stats_in, stats_out, what)
    (540672, 64, 960, 0, 1, 'L1 KPConv contraction'), (540672, 256, 64, 1, 1, 'L1 unary2'), (540672, 256, 128, 0, 1, 'L1 shortcut'),
    (141056, 128, 1920, 0, 1, 'L2 KPConv contraction'), (141056, 128, 512, 0, 1, 'L2 unary1'), (141056, 512, 128, 1, 1, 'L2 unary2'),
    (141056, 512, 256, 0, 1, 'L2 shortcut'), (141056, 64, 960, 0, 1, 'L1->2 strided contraction'),
    (37888, 256, 3840, 0, 1, 'L3 KPConv contraction'), (37888, 256, 1024, 0, 1, 'L3 unary1'), (37888, 1024, 256, 1, 1, 'L3 unary2'),
    (37888, 1024, 512, 0, 1, 'L3 shortcut'), (37888, 128, 1920, 0, 1, 'L2->3 strided contraction'),
    (37888, 768, 256, 0, 0, 'attn in_proj'), (37888, 256, 256, 0, 0, 'attn out_proj'), (37888, 1024, 256, 0, 0, 'FFN 1'),
    (37888, 256, 1024, 0, 0, 'FFN 2'), (227328, 256, 256, 0, 0, 'head MLP (6 layers of tokens)'),
]


def arm():
    from regtr_amd import ops
    dev = 'cuda'
    torch.manual_seed(0)
    out = {}
    for M, N, K, s_in, s_out, what in SHAPES:
        a = torch.randn(M, K, device=dev); w = torch.randn(K, N, device=dev) / K ** 0.5
        sw = ops.SplitWeight(w, 'kn')
        n_clouds = 128
        seg = torch.linspace(0, M, n_clouds + 1, device=dev).to(torch.int32); seg[-1] = M
        max_len = int((seg[1:] - seg[:-1]).max())
        a_stats = ops.instnorm_stats(a, seg, max_len) if s_in else None
        planes = int(os.environ.get('X3_PLANES', '3'))
        if planes != 3: s_in = s_out = 0
        kw = dict(a_stats=a_stats if s_in else None, a_seg_off=seg if s_in else None, want_stats=(seg, max_len) if s_out else None, planes=planes)
        ops.force_x3_gemm = True; ops.use_stream_gemm = False
        def run():
            with ops.f16_pair(ops.f16_pair_default):          # (REGTR_F16_PAIR=0 / 1 arms)
                return ops.gemm(a, sw, **kw)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            r = run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        c = r[0] if isinstance(r, tuple) else r
        rows = torch.randint(0, M, (512,), device=dev)
        ar = a[rows].double()
        if s_in:
            cl = torch.bucketize(rows.int(), seg[1:].contiguous(), right=True)
            u = (ar - a_stats[cl, :, 0].double()) * a_stats[cl, :, 1].double()
            ar = torch.where(u > 0, u, 0.1 * u)
        ref = ar @ w.double()
        err = float((c[rows].double() - ref).abs().max() / ref.abs().max())
        st_err = None
        if s_out:
            st = r[1]
            c0 = c[seg[3]:seg[4]].double()
            st_err = float(max((st[3, :, 0].double() - c0.mean(0)).abs().max(), ((st[3, :, 1].double() - 1 / torch.sqrt(c0.var(0, unbiased=False) + 1e-5)).abs() / st[3, :, 1].double()).max()))
        out[what] = {'M': M, 'N': N, 'K': K, 'us': us, 'TF': 2 * M * N * K / us / 1e6, 'rel_err': err, 'stat_err': st_err}
    print(json.dumps(out))


if __name__ == '__main__':
    if '--arm' in sys.argv:
        arm()
    else:
        res = {}
        arms = (('tiled', {'REGTR_X3_STRIP': '0'}), ('strip', {'REGTR_X3_STRIP': '1'}))
        if '--arms' in sys.argv:      # name=ENV1:val,ENV2:val ...   e.g. --arms "p1=X3_PLANES:1" "noload=REGTR_VARIANT:abl1"
            arms = [(a.split('=')[0], dict(kv.split(':') for kv in a.split('=')[1].split(','))) for a in sys.argv[sys.argv.index('--arms') + 1:]]
        for name, env in arms:
            p = subprocess.run([sys.executable, __file__, '--arm'], capture_output=True, text=True, env=dict(os.environ, **env))
            if p.returncode:
                print(name, 'FAILED', p.stderr[-1500:]); continue
            res[name] = json.loads(p.stdout.strip().splitlines()[-1])
        if '--arms' in sys.argv:
            names = list(res)
            print('| shape | M | N | K | ' + ' | '.join(f'{n} us (TF)' for n in names) + ' |'); print('|---|---|---|---|' + '---|' * len(names))
            for what in res[names[0]]:
                a = res[names[0]][what]
                print(f"| {what} | {a['M']} | {a['N']} | {a['K']} | " + ' | '.join(f"{res[n][what]['us']:.1f} ({res[n][what]['TF']:.0f}) e{res[n][what]['rel_err']:.0e}" for n in names) + ' |')
            print('| sum | | | | ' + ' | '.join(f"{sum(v['us'] for v in res[n].values()):.0f}" for n in names) + ' |')
            sys.exit(0)
        print('| shape | M | N | K | tiled us (TF) | strip us (TF) | x | rel err tiled / strip | stat err strip |'); print('|---|---|---|---|---|---|---|---|---|')
        tot = {'tiled': 0.0, 'strip': 0.0}
        for what in res.get('tiled', {}):
            a, b = res['tiled'][what], res.get('strip', {}).get(what)
            if b is None: continue
            tot['tiled'] += a['us']; tot['strip'] += b['us']
            print(f"| {what} | {a['M']} | {a['N']} | {a['K']} | {a['us']:.1f} ({a['TF']:.0f}) | {b['us']:.1f} ({b['TF']:.0f}) | {a['us'] / b['us']:.2f} | "
                  f"{a['rel_err']:.1e} / {b['rel_err']:.1e} | {b['stat_err'] if b['stat_err'] is None else format(b['stat_err'], '.1e')} |")
        print(f"| sum | | | | {tot['tiled']:.0f} | {tot['strip']:.0f} | {tot['tiled'] / max(tot['strip'], 1e-9):.2f} | | |")
