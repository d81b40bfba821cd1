#!/bin/bash
# Build the C-ABI library of another git revision next to the current one, for same-box A/B runs:
#   tools/ab_lib.sh <rev>                 ->  torchok_amd/lib/libtok_ab.so from <rev> entirely
#   tools/ab_lib.sh <rev> a.hip b.hip ... ->  the WORKING TREE with only the named csrc files taken from <rev> (same C ABI as the
#                                             current library: needed whenever the header gained entry points since <rev>)
#   TOK_LIB=torchok_amd/lib/libtok_ab.so python bench.py ...
set -e
rev=${1:-HEAD}
shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
if [ $# -eq 0 ]; then
  git -C "$root" archive "$rev" torchok_amd/csrc include | tar -x -C "$tmp"
else
  mkdir -p "$tmp/torchok_amd" "$tmp/include"
  cp -r "$root/torchok_amd/csrc" "$tmp/torchok_amd/csrc"
  cp "$root/include/tok.h" "$tmp/include/tok.h"
  for f in "$@"; do git -C "$root" show "$rev:torchok_amd/csrc/$f" > "$tmp/torchok_amd/csrc/$f"; done
fi
objs=()
for f in "$tmp"/torchok_amd/csrc/*.hip "$tmp"/torchok_amd/csrc/*.cpp; do
  [ -e "$f" ] || continue
  o="$tmp/$(basename "$f").o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I"$tmp/include" -c "$f" -o "$o" &
  objs+=("$o")
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/torchok_amd/lib/libtok_ab.so" "${objs[@]}"
rm -rf "$tmp"
echo "built torchok_amd/lib/libtok_ab.so from $rev $*"
