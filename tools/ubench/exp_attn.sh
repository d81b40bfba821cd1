cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -q -k "attn or attention" 2>&1 | grep -E "^E  |passed|failed|assert " | head -12
for f in 0 1; do echo "-- TOK_ATTN_WAVE=$f"; TOK_ATTN_WAVE=$f python bench.py --backbone swinv2_custom --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])"; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/attn_kt -o kt -- python bench.py --backbone swinv2_custom --steps 4 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
db=$(ls gpurun_out/attn_kt/*results.db | head -1); python tools/prof_summary.py $db 7 | grep -i "attn"; rm -rf gpurun_out/attn_kt
