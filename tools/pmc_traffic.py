#!/usr/bin/env python
"""HBM traffic per training step from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md prescribes: they do not fit one pass).

    python tools/pmc_traffic.py <fetch.db> <write.db> <steps_in_run> [out.json] [arena_elements]

Units/corrections (MI355X_MICROARCH.md §HBM): both counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide coalesced streams at 64 B, i.e. reports HALF the bytes -> x2.  The factors are
re-calibrated here on a kernel of known traffic inside the same run (sgd_kernel: 12 B read + 8 B
written per fp32 parameter of the flat arena; adam_kernel: 16 B read + 12 B written) and printed next to the nominal
ones: `implied_fetch_correction` = the factor that makes corrected-fetch / write equal the kernel's known read / write ratio
(1.5 for SGD with momentum, 4/3 for Adam / AdamW), and — when the number of fp32 arena elements of the workload is given —
the absolute factors `fetch_correction_abs`, `write_correction_abs` (known bytes / counted bytes).  Measured in round 4:
WRITE_SIZE is exact (1.000) on both optimizer kernels; FETCH_SIZE x 2 is exact (2.000) on adam_kernel and 12 % short (2.25)
on sgd_kernel, whose gradient operand was written by the weight-gradient kernels just before and is partly served by the
32 MB of L2 (hits never reach the fabric counters) — an undercount of reads that hit L2, not a different unit."""
import json
import sqlite3
import sys


def per_kernel(db_path, counter):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute('select name, count(*), sum(counter_value) from pmc_events where counter_name = ? group by name',
                       (counter,)).fetchall()
    return {n: (c, v) for n, c, v in rows}


def main():
    fetch_db, write_db, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f = per_kernel(fetch_db, 'FETCH_SIZE')
    w = per_kernel(write_db, 'WRITE_SIZE')
    names = sorted(set(f) | set(w), key=lambda n: -(f.get(n, (0, 0))[1] * 2 + w.get(n, (0, 0))[1]))
    tot_f = sum(v for _, v in f.values()) * 1024 / steps
    tot_w = sum(v for _, v in w.values()) * 1024 / steps
    print(f'{"kernel":60s} {"calls/step":>10} {"read MB (x2)":>13} {"write MB":>10}')
    for n in names[:24]:
        c, fv = f.get(n, (0, 0.0))
        _, wv = w.get(n, (0, 0.0))
        print(f'{n[:60]:60s} {c / steps:10.1f} {fv * 2 * 1024 / steps / 1e6:13.1f} {wv * 1024 / steps / 1e6:10.1f}')
    cal = {}
    elems = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    for n in names:
        for tag, rd, wr in (('sgd_kernel', 12, 8), ('adam_kernel', 16, 12)):
            if tag in n and n in f and n in w:
                c, fv = f[n]
                _, wv = w[n]
                key = tag.split('_')[0]
                # per STEP, not per call: an arena with parameters that received no gradient is updated in several runs
                # (SwinV2's unused per-stage norms: two adam_kernel launches per step)
                cal = {f'{key}_calls_per_step': c / steps, f'{key}_fetch_KiB_per_step': fv / steps, f'{key}_write_KiB_per_step': wv / steps,
                       'known_read_over_write': rd / wr, 'implied_fetch_correction': (rd / wr) * wv / fv if fv else None}
                if elems:
                    cal['arena_elements'] = elems
                    cal['fetch_correction_abs'] = elems * rd / (fv / steps * 1024)
                    cal['write_correction_abs'] = elems * wr / (wv / steps * 1024)
    res = {'fetch_bytes_per_step_raw': tot_f, 'write_bytes_per_step_raw': tot_w,
           'fetch_correction': 2.0, 'write_correction': 1.0,
           'hbm_bytes_per_step': tot_f * 2.0 + tot_w, 'steps_in_run': steps, 'calibration': cal}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 4:
        json.dump(res, open(sys.argv[4], 'w'), indent=1)


if __name__ == '__main__':
    main()
