#!/bin/bash
# conv_win_kernel with parts of a stage removed (TOK_WIN_PROBE masks, conv_win.hip: results invalid, timing only):
#   build here:   bash tools/ubench/win_probe.sh build      (probe libraries under torchok_amd/lib/probe/, they travel with gpurun)
#   on the box:   bash tools/ubench/win_probe.sh run        (one 3x3 layer per library, tools/ubench/one_conv_time.py)
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
P=torchok_amd/lib/probe
if [ "$1" = build ]; then
  mkdir -p $P
  for m in 1 2 3 4 8 16 32; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DTOK_WIN_PROBE=$m -c torchok_amd/csrc/conv_win.hip -o $P/conv_win_$m.o || exit 1
    objs=$(ls torchok_amd/lib/obj/*.o | grep -v conv_win.hip.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $P/libtok_probe_$m.so $objs $P/conv_win_$m.o || exit 1
    rm $P/conv_win_$m.o
  done
  ls -la $P
else
  for shape in "12 256 48 48" "12 128 96 96"; do
    echo "== (n h c k) $shape  3x3 forward, us per call"
    printf "%-44s" "full kernel"; python tools/ubench/one_conv_time.py $shape 3 1 fwd
    for m in 1 2 3 4 8 16 32; do
      case $m in 1) d="no window-fragment reads after tap 0";; 2) d="no weight-fragment reads after tap 0";; 3) d="neither";; 4) d="no block barrier";; 8) d="no DMA (weights + window)";; 16) d="no MFMAs";; 32) d="no epilogue";; esac
      printf "%-44s" "mask $m: $d"; TOK_LIB=$PWD/$P/libtok_probe_$m.so python tools/ubench/one_conv_time.py $shape 3 1 fwd
    done
  done
fi
