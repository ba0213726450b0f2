#!/bin/bash
# Collects what profiles/<round>/ keeps: kernel-trace statistics, the two HBM traffic counter passes and two
# SQ counter passes, each in a run of its own, for the decompress and the compress bench.
# Run on the GPU box from the repo root; writes under gpurun_out/prof/ (tools/collect_profiles.py condenses it).
set -o pipefail
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
for op in decompress compress; do
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${op}_stats" -- \
      python3 "$ROOT/bench.py" --op $op --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > "$OUT/${op}_bench_line.json" || exit 1
  echo "[profile] $op kernel stats done"
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" \
              "sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE" \
              "sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_INSTS_FLAT"; do
    set -- $pass; tag=$1; shift
    timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/${op}_$tag" -- \
        python3 "$ROOT/bench.py" --op $op --steps 1 --warmup 0 --no-cpu-baseline --no-secondary > /dev/null || exit 1
    echo "[profile] $op $tag done"
  done
done
