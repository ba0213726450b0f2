#!/bin/bash
# Collects what profiles/<round>/ keeps: kernel-trace statistics, the two HBM traffic counter passes and two
# SQ counter passes, each in a run of its own, for the decompress and the compress bench.
# The other per-GPU shapes of the bench (8192 x 64 KiB gzip decompress, 128 x 4 MiB JSON level-12 compress) get
# kernel-trace statistics only.
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
for shape in "decompress64k --op decompress --streams 8192 --buf-kib 64 --steps 3 --warmup 1" \
             "compressl12 --op compress --streams 128 --buf-kib 4096 --corpus json --level 12 --distinct 16 --steps 2 --warmup 1" \
             "compressl1 --op compress --level 1 --steps 2 --warmup 1"; do
  set -- $shape; tag=$1; shift
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${tag}_stats" -- \
      python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --no-secondary > "$OUT/${tag}_bench_line.json" || exit 1
  echo "[profile] $tag kernel stats done"
done
