#!/bin/bash
# Per-kernel durations of a short bench run: tools/kstats.sh <tag> [bench args...] (GPU box, from the repo root)
ROOT="$(pwd)"; tag="$1"; shift
OUT="$ROOT/gpurun_out/kstats/$tag"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-secondary --steps 3 --warmup 1 "$@" > "$OUT/bench_line.json" || exit 1
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-60s calls %4s avg_us %10.1f total_ms %9.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
