#!/usr/bin/env python
"""Timeline view of ONE training step out of a rocprofv3 --kernel-trace rocpd database:
per-queue busy time, idle gaps on the main queue, how much of the side queue's work overlaps the main queue,
and the per-kernel time of the main queue in dispatch order (optionally dumped).

    python tools/timeline.py <results.db> [--step -2] [--dump]

A step is delimited by consecutive `sgd_kernel` / `adam_kernel` dispatches (the optimizer step is one launch)."""
import argparse
import sqlite3
from collections import defaultdict


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    for k in ('conv_igemm_kernel', 'conv_wgrad_kernel'):
        if k in n:
            return n[n.index(k):n.index('(', n.index(k))] if '(' in n[n.index(k):] else n[n.index(k):]
    if n.startswith('_ZN12_GLOBAL__N_1'):
        import re
        m = re.match(r'_ZN12_GLOBAL__N_1(\d+)', n)
        if m:
            L = int(m.group(1))
            s = n[len(m.group(0)):]
            return s[:L]
    return n.split('(')[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--step', type=int, default=-2)
    ap.add_argument('--dump', action='store_true')
    ap.add_argument('--opt', default='sgd_kernel,adam_kernel')
    a = ap.parse_args()
    cur = sqlite3.connect(a.db).cursor()
    rows = cur.execute('select name, start, end, queue_id, stream_id, grid_x, workgroup_x from kernels order by start').fetchall()
    marks = [i for i, r in enumerate(rows) if any(o in r[0] for o in a.opt.split(','))]
    if len(marks) < 3:
        raise SystemExit('fewer than 3 optimizer launches in the trace')
    lo, hi = marks[a.step - 1] + 1, marks[a.step] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print(f'step: {len(step)} dispatches, wall {(t1 - t0) / 1e6:.3f} ms')
    byq = defaultdict(list)
    for r in step:
        byq[(r[3], r[4])].append(r)
    main_q = max(byq, key=lambda q: sum(r[2] - r[1] for r in byq[q]))     # the dependent chain: most busy time
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r[2] - r[1] for r in rs)
        print(f'  queue {q}: {len(rs):4d} dispatches, busy {busy / 1e6:7.3f} ms, span {(rs[0][1] - t0) / 1e6:.2f}..{(max(r[2] for r in rs) - t0) / 1e6:.2f} ms'
              + ('   <- main' if q == main_q else ''))
    mq = byq[main_q]
    gaps = [(mq[i + 1][1] - mq[i][2]) for i in range(len(mq) - 1)]
    gaps_pos = [g for g in gaps if g > 0]
    print(f'main queue: sum of gaps {sum(gaps_pos) / 1e6:.3f} ms over {len(gaps_pos)} gaps; '
          f'gaps > 5 us: {sum(1 for g in gaps_pos if g > 5000)} totalling {sum(g for g in gaps_pos if g > 5000) / 1e6:.3f} ms')
    # union of side-queue intervals and its overlap with main busy intervals
    side = sorted((r[1], r[2]) for q, rs in byq.items() if q != main_q for r in rs)
    def union(iv):
        out = []
        for s, e in iv:
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out
    su = union(side)
    mu = union(sorted((r[1], r[2]) for r in mq))
    def inter(A, B):
        i = j = 0
        t = 0
        while i < len(A) and j < len(B):
            s, e = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
            if e > s:
                t += e - s
            if A[i][1] < B[j][1]:
                i += 1
            else:
                j += 1
        return t
    print(f'side queues busy (union) {sum(e - s for s, e in su) / 1e6:.3f} ms, of which overlapping main kernels {inter(su, mu) / 1e6:.3f} ms')
    # main queue per kernel class, alone vs overlapped
    agg = defaultdict(lambda: [0, 0, 0])
    for r in mq:
        ov = inter([[r[1], r[2]]], su)
        k = short(r[0])
        agg[k][0] += 1
        agg[k][1] += r[2] - r[1]
        agg[k][2] += ov
    print(f'{"main-queue kernel":60s} {"calls":>5} {"ms":>8} {"ms overlapped by side":>22}')
    for k, (n, t, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k[:60]:60s} {n:5d} {t / 1e6:8.3f} {ov / 1e6:22.3f}')
    agg = defaultdict(lambda: [0, 0])
    for q, rs in byq.items():
        if q == main_q:
            continue
        for r in rs:
            k = short(r[0])
            agg[k][0] += 1
            agg[k][1] += r[2] - r[1]
    print(f'{"side-queue kernel":60s} {"calls":>5} {"ms":>8}')
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'{k[:60]:60s} {n:5d} {t / 1e6:8.3f}')
    if a.dump:
        for r in step:
            print(f'{(r[1] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:8.1f} q{r[3]} g{r[5]:>7} {short(r[0])}')


if __name__ == '__main__':
    main()
