#!/bin/bash
# Runs the bench once per tuning variant under variants/ (see tools/build_variant.sh) on the GPU box:
# tools/sweep_variants.sh [bench args...]; one JSON line per variant in gpurun_out/sweep/<name>.json.
# Stops at the first failing run.
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/sweep"
mkdir -p "$OUT"
for so in "$ROOT"/variants/*/librsx_deflate.so; do
  name="$(basename "$(dirname "$so")")"
  echo "[sweep] $name"
  RSX_LIBRARY="$so" timeout -k 10 240 python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  rc=$?
  if [ $rc -ne 0 ]; then echo "[sweep] $name failed rc=$rc"; tail -5 "$OUT/$name.err"; exit $rc; fi
  python3 -c "import json,sys; r=json.load(open('$OUT/$name.json')); print('[sweep] %s kernel_ms %.3f value %.1f' % ('$name', r['roofline']['kernel_ms'], r['value']))"
done
