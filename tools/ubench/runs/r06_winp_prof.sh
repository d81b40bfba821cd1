#!/bin/bash
# kernel-level split of the 3x3 weight gradients (window kernel vs slab reduce)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_winp_prof; mkdir -p $O
export TMPDIR=/tmp
for net in resnet50 hrnet_w48; do
  B=256; [ $net = hrnet_w48 ] && B=24
  rocprofv3 --kernel-trace --stats -d $O/$net -o p -- python tools/bench_conv.py --what wgrad --net $net --batch $B > $O/$net.log 2>&1
  f=$(find $O/$net -name '*kernel_stats.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
done
