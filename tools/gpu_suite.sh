#!/bin/bash
# Full GPU test suite + the default bench line (what the driver runs at round end):  tools/gpu_suite.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=${1:-gpurun_out/suite}; mkdir -p $o
rm -f gpurun_out/parity_distances.jsonl
timeout 1800 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1; tail -4 $o/gpu_tests.txt
# the parity distances the suite just measured, folded into one record of THIS tree (commit it as profiles/rNN_parity_distances.json)
python tools/parity_record.py gpurun_out/parity_distances.jsonl $o/parity_distances.json > $o/parity_record.txt 2>&1; tail -2 $o/parity_record.txt
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
python bench.py > $o/bench.json 2> $o/bench.err; cut -c1-2400 $o/bench.json
