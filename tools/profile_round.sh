#!/bin/bash
# Collects the rocprofv3 summaries kept under profiles/: kernel-trace statistics and the
# two HBM-traffic counter passes (each in a run of its own) for the decompress and the
# compress bench.  Run on the GPU box from the repo root; writes under gpurun_out/prof/.
set -e -o pipefail
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
for op in decompress compress; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${op}_stats" -- \
      python3 "$ROOT/bench.py" --op $op --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/${op}_bench_line.json"
  echo "[profile] $op kernel stats done"
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${op}_fetch" -- \
      python3 "$ROOT/bench.py" --op $op --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null
  echo "[profile] $op FETCH_SIZE done"
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${op}_write" -- \
      python3 "$ROOT/bench.py" --op $op --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null
  echo "[profile] $op WRITE_SIZE done"
done
