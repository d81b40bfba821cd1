#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db, the ROCm 7.2 default output) into the
per-kernel table committed under profiles/:   python tools/prof_summary.py <results.db> [steps]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cur = db.cursor()
    rows = cur.execute('select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) '
                       'from kernels group by name order by 3 desc').fetchall()
    tot = sum(r[2] for r in rows)
    print(f'# total kernel time {tot / 1e6:.3f} ms over {steps} step(s) = {tot / 1e6 / steps:.3f} ms/step')
    print(f'{"ms/step":>9} {"%":>6} {"calls/step":>10} {"avg_us":>9} {"min_us":>9} {"max_us":>9}  kernel')
    for name, n, s, a, mn, mx in rows:
        print(f'{s / 1e6 / steps:9.3f} {100 * s / tot:6.2f} {n / steps:10.1f} {a / 1e3:9.1f} {mn / 1e3:9.1f} '
              f'{mx / 1e3:9.1f}  {name[:120]}')


if __name__ == '__main__':
    main()
