"""One forward's kernel sequence from a rocprofv3 rocpd database: order, short name, grid, duration.
usage: python tools/trace_forward.py <results.db> [n_last_kernels]   (prints the launches of the LAST forward in the trace)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:64]


def main(db, marker='k_procrustes'):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    gcols = [c for c in ('grid_size_x', 'grid_x', 'grid_size') if c in cols]
    wcols = [c for c in ('workgroup_size_x', 'workgroup_x', 'workgroup_size') if c in cols]
    sel = f'{name_col}, start, end' + (f', {gcols[0]}' if gcols else ', 0') + (f', {wcols[0]}' if wcols else ', 0')
    rows = cur.execute(f'select {sel} from kernels order by start').fetchall()
    ends = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(ends) < 2:
        lo, hi = 0, len(rows)
    else:
        lo, hi = ends[-2] + 1, ends[-1] + 1
    t0 = rows[lo][1]
    tot = 0
    print('| # | start us | dur us | kernel | grid | wg |')
    print('|---|---|---|---|---|---|')
    for i, (n, s, e, g, w) in enumerate(rows[lo:hi]):
        tot += e - s
        print(f'| {i} | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {short(n)} | {g} | {w} |')
    print(f'| | wall {(rows[hi - 1][2] - t0) / 1e3:.1f} | busy {tot / 1e3:.1f} | {hi - lo} launches | | |')


if __name__ == '__main__':
    main(*sys.argv[1:3])
