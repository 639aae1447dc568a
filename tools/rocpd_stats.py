"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into the familiar --stats table.
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'void ', '', name)
    m = re.match(r'([\w:<>, ]+?)\(', name)
    return (m.group(1) if m else name)[:70]


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f'select {name_col}, start, end from kernels').fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 10 ** 18, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'| {k} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / tot:.1f} |')
    lines.append(f'| TOTAL | {len(rows)} | {tot / 1e6:.3f} | | | | 100 |')
    txt = '\n'.join(lines)
    print(txt)
    if out:
        open(out, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(*sys.argv[1:3])
