#!/usr/bin/env python
"""Per-queue kernel sequence inside a time window of one training step of a rocprofv3 --kernel-trace database.

    python tools/queue_window.py <results.db> --from-ms 14 --to-ms 20 [--step -2]
One line per queue and 100-us bucket would hide what matters here (who STARTS when), so every kernel is listed: start, end (ms
from the step's first kernel), queue, short name."""
import argparse
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from timeline import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--step', type=int, default=-2)
    ap.add_argument('--from-ms', type=float, default=0.0)
    ap.add_argument('--to-ms', type=float, default=5.0)
    ap.add_argument('--opt', default='sgd_kernel,adam_kernel')
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute('select name, start, end, queue_id, stream_id from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if any(o in r[0] for o in a.opt.split(','))]
    lo, hi = marks[a.step - 1] + 1, marks[a.step] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    qs = sorted({(r[3], r[4]) for r in step})
    col = {q: i for i, q in enumerate(qs)}
    print('queues:', qs)
    for r in step:
        s, e = (r[1] - t0) / 1e6, (r[2] - t0) / 1e6
        if e < a.from_ms or s > a.to_ms:
            continue
        print(f'{s:8.3f} {e:8.3f}  ' + '    ' * col[(r[3], r[4])] + f'q{col[(r[3], r[4])]} {short(r[0])[:44]}')


if __name__ == '__main__':
    main()
