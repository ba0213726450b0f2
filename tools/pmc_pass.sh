#!/bin/bash
# One rocprofv3 counter pass over a short bench run: tools/pmc_pass.sh <op> <tag> <counters...>
# Run on the GPU box from the repo root; writes gpurun_out/pmc/<tag>/.
set -e -o pipefail
ROOT="$(pwd)"
op="$1"; tag="$2"; shift 2
OUT="$ROOT/gpurun_out/pmc/$tag"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT" -- \
    python3 "$ROOT/bench.py" --op "$op" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null
echo "[pmc] $tag done"
