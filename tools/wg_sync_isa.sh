#!/bin/bash
# Compiles the three kernel files to gfx950 ISA twice (WG_SYNC as shipped, and with plain
# __syncthreads() via -DRSX_PLAIN_SYNC=1) and lists every barrier a wave can reach with an LDS
# access still outstanding.  Needs no GPU.   usage: tools/wg_sync_isa.sh [outdir]
set -e
cd "$(dirname "$0")/.."
out=${1:-/tmp/wg_sync_isa}
mkdir -p "$out"
for f in inflate deflate checksum; do
  for mode in wgsync plain; do
    def=""; [ $mode = plain ] && def="-DRSX_PLAIN_SYNC=1"
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 $def -x hip -S --cuda-device-only \
        libdeflate-rsx_amd/csrc/$f.hip -o "$out/${f}_$mode.s" 2>/dev/null
    echo "== $f.hip, $mode"
    python3 tools/barrier_waits.py "$out/${f}_$mode.s"
  done
done
