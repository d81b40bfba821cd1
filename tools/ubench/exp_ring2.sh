cd $GRAFT_REPO_ROOT
for lib in torchok_amd/lib/libtok_gfx950.so torchok_amd/lib/libtok_fake1.so torchok_amd/lib/libtok_fake0.so; do echo "-- $lib"; TOK_CONV_RING_MIN_TILES=1 python tools/bench_conv.py --lib $lib --what fwd,dgrad 2>&1 | grep ", 3, 1)"; done
