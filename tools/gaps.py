#!/usr/bin/env python
"""Where the main queue idles in one training step of a rocprofv3 --kernel-trace database: every gap above a threshold with the
kernel in front of it and behind it, what the other queues were running meanwhile, and totals per (before -> after) pair.

    python tools/gaps.py <results.db> [--step -2] [--min-us 8] [--top 40]"""
import argparse
import sqlite3
import sys
import os
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from timeline import short  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--step', type=int, default=-2)
    ap.add_argument('--min-us', type=float, default=8.0)
    ap.add_argument('--top', type=int, default=40)
    ap.add_argument('--opt', default='sgd_kernel,adam_kernel')
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute('select name, start, end, queue_id, stream_id, grid_x, workgroup_x from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if any(o in r[0] for o in a.opt.split(','))]
    lo, hi = marks[a.step - 1] + 1, marks[a.step] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    byq = defaultdict(list)
    for r in step:
        byq[(r[3], r[4])].append(r)
    main_q = max(byq, key=lambda q: sum(r[2] - r[1] for r in byq[q]))
    mq = byq[main_q]
    others = [r for q, rs in byq.items() if q != main_q for r in rs]
    pairs = defaultdict(lambda: [0, 0.0])
    gaps = []
    for i in range(len(mq) - 1):
        g = mq[i + 1][1] - mq[i][2]
        if g > a.min_us * 1e3:
            s, e = mq[i][2], mq[i + 1][1]
            busy = [o for o in others if o[1] < e and o[2] > s]
            cover = sum(min(o[2], e) - max(o[1], s) for o in busy)
            gaps.append((g, s - t0, short(mq[i][0])[:40], short(mq[i + 1][0])[:40], len(busy), cover,
                         ', '.join(sorted({short(o[0])[:28] for o in busy}))[:90]))
            k = (short(mq[i][0])[:40], short(mq[i + 1][0])[:40])
            pairs[k][0] += 1
            pairs[k][1] += g
    tot = sum(g[0] for g in gaps)
    print(f'main queue {main_q}: {len(mq)} dispatches; {len(gaps)} gaps > {a.min_us} us totalling {tot / 1e6:.3f} ms')
    print('--- by (kernel before -> kernel after)')
    for k, (n, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print(f'{t / 1e3:9.1f} us {n:4d}x  {k[0]} -> {k[1]}')
    print('--- largest gaps: us, at ms, before -> after | other-queue kernels running in the gap (count, covered us)')
    for g in sorted(gaps, reverse=True)[:a.top]:
        print(f'{g[0] / 1e3:8.1f} @{g[1] / 1e6:7.2f}  {g[2]} -> {g[3]} | {g[4]} ({g[5] / 1e3:.0f} us): {g[6]}')


if __name__ == '__main__':
    main()
