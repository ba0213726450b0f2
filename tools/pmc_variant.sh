#!/bin/bash
# usage: pmc_variant.sh <variant> <counter>
set -o pipefail
ROOT="$(pwd)"; v=$1; c=$2
OUT="$ROOT/gpurun_out/pmc/${v}_$c"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
export RSX_LIBRARY="$ROOT/variants/$v/librsx_deflate.so"
timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --op compress --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null 2>&1
echo "[pmc] $v $c done"
