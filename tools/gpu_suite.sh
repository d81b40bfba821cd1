#!/bin/bash
# Full GPU test suite + the default bench line (what the driver runs at round end):  tools/gpu_suite.sh <outdir>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
o=${1:-gpurun_out/suite}; mkdir -p $o
timeout 1500 python -m pytest tests -m gpu -x -q > $o/gpu_tests.txt 2>&1; tail -4 $o/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > $o/smoke.txt 2>&1; tail -2 $o/smoke.txt
python bench.py > $o/bench.json 2> $o/bench.err; cut -c1-2400 $o/bench.json
