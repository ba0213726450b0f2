#!/bin/bash
# Builds a tuning variant of the engine library: tools/build_variant.sh <name> <extra hipcc flags...>
# -> variants/<name>/librsx_deflate.so (select it at run time with RSX_LIBRARY=<path>).
set -e
name="$1"; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/libdeflate-rsx_amd/csrc"
B="$ROOT/variants/$name"
mkdir -p "$B"
pids=()
for f in api.cpp checksum.hip inflate.hip deflate.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -x hip -c "$SRC/$f" -o "$B/$f.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p" || { rm -rf "$B"; echo "build of variant $name failed"; exit 1; }; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$B/librsx_deflate.so" "$B"/*.o
rm -f "$B"/*.o
echo "built $B/librsx_deflate.so"
