"""GPU busy fraction of a run from a rocprofv3 rocpd database: the UNION of the kernels' [start, end) intervals over the window of the last
`procrustes` launches (the timed forwards), against the window's wall time -- with forwards in flight on several streams the per-kernel sums say
nothing about idle time, the union does.   usage: python tools/busy_union.py <results.db> [n_last_forwards=12]"""
import sqlite3
import sys


def main(db, n_last=12):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f'select {name_col}, start, end from kernels order by start').fetchall()
    marks = [r[2] for r in rows if 'k_procrustes' in r[0]]
    n_last = min(int(n_last), len(marks) - 1)
    t_lo, t_hi = marks[-n_last - 1], marks[-1]
    iv = sorted((max(s, t_lo), min(e, t_hi)) for _, s, e in rows if e > t_lo and s < t_hi)
    busy, cur_s, cur_e, gaps = 0, None, None, []
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    wall = t_hi - t_lo
    gaps.sort(reverse=True)
    print(f'window: last {n_last} forwards, {wall / 1e6:.2f} ms; some kernel running {busy / 1e6:.2f} ms = {100 * busy / wall:.1f} % of it; '
          f'{len(gaps)} idle gaps, total {sum(gaps) / 1e6:.2f} ms, largest {[round(g / 1e3, 1) for g in gaps[:8]]} us; '
          f'sum of kernel durations {sum(e - s for s, e in iv) / 1e6:.2f} ms ({sum(e - s for s, e in iv) / wall:.2f} x the window)')


if __name__ == '__main__':
    main(*sys.argv[1:3])
