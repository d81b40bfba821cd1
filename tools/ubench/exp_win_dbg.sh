cd $GRAFT_REPO_ROOT
for c in 17 18 19 20 21; do echo "== case $c"; timeout 120 python -m pytest tests/test_kernels_gpu.py -q -x -k "test_conv_fwd and case$c" 2>&1 | grep -v "^  File\|^$\|Extension\|Current thread" | tail -4; done
