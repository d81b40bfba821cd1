#!/usr/bin/env python
"""MFMA utilisation of a training step from a rocprofv3 PMC pass

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python bench.py ...
    python tools/pmc_mfma.py <results.db> <steps_in_run> [out.json]

SQ_VALU_MFMA_BUSY_CYCLES counts cycles in which a SIMD's matrix pipe is busy, summed over the SIMDs the counter instance
covers (MI355X_MICROARCH.md: = 32 x N_mfma for 32x32x16 bf16, i.e. issue-paced pipe time); GRBM_GUI_ACTIVE counts the cycles
the dispatch kept the GPU busy (reported once per XCC: averaged here).  Check: the token GEMMs of SwinV2-T come out at 0.18,
their FLOP rate is 18 % of the dense bf16 peak.  Utilisation of a dispatch = MFMA busy cycles / (GRBM_GUI_ACTIVE x SIMDs), SIMDs = 256 CUs x 4.
Reported per kernel (time-weighted by GRBM_GUI_ACTIVE) and for the whole step."""
import json
import sqlite3
import sys

SIMDS = 256 * 4


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    cur = sqlite3.connect(db).cursor()
    names = [r[0] for r in cur.execute('select distinct counter_name from pmc_events').fetchall()]
    # GRBM_GUI_ACTIVE comes as one value per XCC (8 per dispatch, each ~ the dispatch's active cycles): averaged.
    # SQ_VALU_MFMA_BUSY_CYCLES comes per shader-engine instance: summed (busy cycles of all 1024 SIMDs).
    rows = cur.execute('select name, dispatch_id, counter_name, sum(counter_value), count(*) from pmc_events '
                       'group by dispatch_id, counter_name').fetchall()
    per = {}
    for name, did, cn, v, n in rows:
        per.setdefault(did, {'name': name})[cn] = v / n if cn == 'GRBM_GUI_ACTIVE' else v
    agg = {}
    tot_busy = tot_act = 0.0
    for d in per.values():
        busy, act = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), d.get('GRBM_GUI_ACTIVE', 0.0)
        a = agg.setdefault(d['name'], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += busy
        a[2] += act
        tot_busy += busy
        tot_act += act
    print(f'# counters in the pass: {names}')
    print(f'{"kernel":70s} {"calls/step":>10} {"GUI_ACTIVE Mcyc/step":>21} {"MFMA util":>10}')
    for n, (c, b, a) in sorted(agg.items(), key=lambda kv: -kv[1][2])[:24]:
        print(f'{n[:70]:70s} {c / steps:10.1f} {a / steps / 1e6:21.3f} {b / (a * SIMDS) if a else 0:10.4f}')
    util = tot_busy / (tot_act * SIMDS) if tot_act else 0.0
    res = {'mfma_busy_cycles_per_step': tot_busy / steps, 'gui_active_cycles_per_step': tot_act / steps, 'simds': SIMDS,
           'mfma_util_busy_over_active': util, 'steps_in_run': steps,
           'note': 'sum over dispatches of SQ_VALU_MFMA_BUSY_CYCLES / (sum of per-XCC-averaged GRBM_GUI_ACTIVE x 1024 SIMDs); dispatches that '
                   'overlap on two streams each count their own active cycles'}
    print(json.dumps(res, indent=1))
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], 'w'), indent=1)


if __name__ == '__main__':
    main()
