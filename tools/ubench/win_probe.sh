#!/bin/bash
# Probe builds of conv_win.hip (TOK_WIN_PROBE masks): libtok_winprobe_<mask>.so next to the library; run on the GPU box with
#   for m in ...; do python tools/bench_conv.py --lib torchok_amd/lib/libtok_winprobe_$m.so --what fwd --net hrnet_w48 --batch 24; done
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
objs=$(ls $root/torchok_amd/lib/obj/*.o | grep -v conv_win)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DTOK_WIN_PROBE=$m -c $root/torchok_amd/csrc/conv_win.hip -o /tmp/conv_win_probe_$m.o &
done
wait
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/torchok_amd/lib/libtok_winprobe_$m.so $objs /tmp/conv_win_probe_$m.o
done
ls -la $root/torchok_amd/lib/libtok_winprobe_*.so
