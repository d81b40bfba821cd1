cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for lib in ${LIBS:-torchok_amd/lib/libtok_gfx950.so}; do echo "-- $lib"
TOK_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/attn_kt -o kt -- python bench.py --backbone swinv2_custom --steps 4 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
db=$(ls gpurun_out/attn_kt/*results.db | head -1); python tools/prof_summary.py $db 7 | grep -i "attn"; rm -rf gpurun_out/attn_kt
done
