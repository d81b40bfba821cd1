#!/bin/bash
# which (hardware queue, stream) torch's ProcessGroupNCCL runs a collective on: tools/ubench/nccl_stream_probe.py under a kernel trace
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=gpurun_out/nccl_probe; mkdir -p $o
timeout 300 rocprofv3 --kernel-trace -d $o/raw -o kt -- python tools/ubench/nccl_stream_probe.py > $o/log.txt 2>&1
tail -2 $o/log.txt
db=$(ls $o/raw/*results.db | head -1)
python - "$db" <<'PY' | tee $o/queues.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute('select name, queue_id, stream_id from kernels order by start').fetchall()
keep = [r for r in rows if 'spin' not in r[0]]
for r in keep[-20:]:
    print(f'queue {r[1]} stream {r[2]}  {r[0][:90]}')
PY
rm -rf $o/raw
