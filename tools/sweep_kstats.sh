#!/bin/bash
# Per-kernel durations for every tuning variant under variants/: tools/sweep_kstats.sh [bench args] (GPU box)
ROOT="$(pwd)"
for so in "$ROOT"/variants/*/librsx_deflate.so; do
  name="$(basename "$(dirname "$so")")"
  echo "[kstats] $name"
  RSX_LIBRARY="$so" tools/kstats.sh "$name" "$@" | grep -E "inflate_|deflate_" || exit 1
done
