"""Summarise rocprofv3 --pmc csv output (one directory per counter pass) into a per-kernel table:
mean counter value per dispatch.  usage: python tools/pmc_summary.py <outdir> [kernel-substring]"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:60]


def main(out, filt=None):
    vals = defaultdict(lambda: defaultdict(list))    # kernel -> counter -> [values]
    for f in sorted(glob.glob(f'{out}/pmc_*/**/*counter_collection.csv', recursive=True)):
        per_dispatch = defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            key = (row['Dispatch_Id'], row['Counter_Name'])
            per_dispatch[key] += float(row['Counter_Value'])
            names[row['Dispatch_Id']] = short(row['Kernel_Name'])
        for (d, c), v in per_dispatch.items():
            vals[names[d]][c].append(v)
    counters = sorted({c for k in vals for c in vals[k]})
    print('| kernel | dispatches | ' + ' | '.join(counters) + ' |')
    print('|---|---|' + '---|' * len(counters))
    for k in sorted(vals, key=lambda k: -sum(vals[k].get('SQ_BUSY_CYCLES', vals[k].get('GRBM_GUI_ACTIVE', [0])))):
        if filt and filt not in k:
            continue
        n = max(len(v) for v in vals[k].values())
        cells = []
        for c in counters:
            v = vals[k].get(c)
            cells.append(f'{sum(v) / len(v):.4g}' if v else '')
        print(f'| {k} | {n} | ' + ' | '.join(cells) + ' |')


def traffic_json(out, kernel_sub, calib_sub='k_instnorm_partial'):
    """HBM bytes per launch of the kernels matching `kernel_sub`, from the FETCH_SIZE / WRITE_SIZE passes (KB units).
    FETCH_SIZE on gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): doubled here."""
    import json
    vals = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(f'{out}/pmc_*/**/*counter_collection.csv', recursive=True)):
        per = defaultdict(float); names = {}
        for row in csv.DictReader(open(f)):
            if row['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                per[(row['Dispatch_Id'], row['Counter_Name'])] += float(row['Counter_Value'])
                names[row['Dispatch_Id']] = short(row['Kernel_Name'])
        for (d, c), v in per.items():
            vals[names[d]][c].append(v)
    res = {}
    for k in vals:
        if kernel_sub in k or calib_sub in k:
            f, w = vals[k].get('FETCH_SIZE', []), vals[k].get('WRITE_SIZE', [])
            res[k] = {'launches': len(f), 'fetch_bytes_per_launch': 2 * 1024 * sum(f) / max(len(f), 1),
                      'write_bytes_per_launch': 1024 * sum(w) / max(len(w), 1),
                      'fetch_bytes_total': 2 * 1024 * sum(f), 'write_bytes_total': 1024 * sum(w)}
    print(json.dumps(res, indent=1))


def bench_traffic_json(out, tag, pairs=64, points=20000, shuffle=False):
    """profiles/pmc_traffic.json for bench.py's `roofline.traffic`: mean HBM bytes per KPConv-gather launch of the PMC passes under
    `out` (taken over `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline` on the default workload)."""
    import io, json, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        traffic_json(out, 'k_kpconv_gather')
    kernels = json.loads(buf.getvalue())
    g = {k: v for k, v in kernels.items() if 'k_kpconv_gather' in k}
    n = sum(v['launches'] for v in g.values())
    total = sum(v['fetch_bytes_total'] + v['write_bytes_total'] for v in g.values())
    print(json.dumps({'workload': {'pairs': pairs, 'points': points, 'shuffle': shuffle}, 'hbm_bytes_per_launch': total / max(n, 1),
                      'source': f'profiles/{tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel-trace only) over '
                                '`python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline`; bytes = 2 x FETCH_SIZE KB (gfx950 tallies 128-B '
                                f'requests at 64 B) + WRITE_SIZE KB; mean over the {n} gather launches of {n // 11} forwards',
                      'kernels': kernels}, indent=1))


def mfma_table(out):
    """Matrix-core utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles =
    GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs); plus the wave-cycle split (SQ_* are quad-cycles)."""
    vals = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
    for f in sorted(glob.glob(f'{out}/pmc_*/**/*counter_collection.csv', recursive=True)):
        seen = set()
        for row in csv.DictReader(open(f)):
            k = short(row['Kernel_Name'])
            vals[k][row['Counter_Name']] += float(row['Counter_Value'])
            if row['Counter_Name'] == 'GRBM_GUI_ACTIVE' and row['Dispatch_Id'] not in seen:
                seen.add(row['Dispatch_Id']); cnt[k] += 1
    print('| kernel | launches | MFMA busy % of SIMD-cycles | wave cycles: waitcnt % | issue stall % | VALU % | MFMA instr / launch |')
    print('|---|---|---|---|---|---|---|')
    for k in sorted(vals, key=lambda k: -vals[k].get('GRBM_GUI_ACTIVE', 0)):
        v = vals[k]
        if not v.get('GRBM_GUI_ACTIVE') or not v.get('SQ_WAVE_CYCLES'):
            continue
        util = 100 * v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)
        wc = v['SQ_WAVE_CYCLES']
        print(f"| {k} | {cnt[k]} | {util:.1f} | {100 * v.get('SQ_WAIT_ANY', 0) / wc:.0f} | {100 * v.get('SQ_WAIT_INST_ANY', 0) / wc:.0f} | "
              f"{100 * v.get('SQ_ACTIVE_INST_VALU', 0) / wc:.0f} | {v.get('SQ_INSTS_MFMA', 0) / max(cnt[k], 1):.3g} |")


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--mfma':
        mfma_table(sys.argv[2])
    elif len(sys.argv) > 3 and sys.argv[1] == '--bench-traffic':
        bench_traffic_json(sys.argv[2], sys.argv[3])
    elif len(sys.argv) > 2 and sys.argv[1] == '--traffic':
        traffic_json(*sys.argv[2:4])
    else:
        main(*sys.argv[1:3])
