"""Copies the newest summaries of a tools/profile_round.sh + tools/pmc_pass.sh run from gpurun_out/ into
profiles/<round>/ and rewrites bench_traffic.json from the two HBM counter passes.
usage: python tools/collect_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
base, dest = "gpurun_out/prof", "profiles/" + rnd
os.makedirs(dest, exist_ok=True)


def newest(pattern):
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1]


def kernel_stats(op):
    f = newest("%s/%s_stats/**/*kernel_stats.csv" % (base, op))
    for r in list(csv.DictReader(open(f)))[:3]:
        print(op, r["Name"][:56], r["Calls"], "%.3f ms" % (float(r["AverageNs"]) / 1e6))
    shutil.copy(f, "%s/%s_kernel_stats.csv" % (dest, op))


def counter_kb(op, kind, name, kernel):
    f = newest("%s/%s_%s/**/*counter_collection.csv" % (base, op, kind))
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
    shutil.copy(f, "%s/%s_pmc_%s.csv" % (dest, op, name))
    print(op, name, "%.0f KB per launch over %d launches" % (sum(vals) / len(vals), len(vals)))
    return sum(vals) / len(vals) * 1024


total = {}
for op, kernel in (("decompress", "inflate_kernel"), ("compress", "deflate_chunks_kernel")):
    kernel_stats(op)
    fetch = counter_kb(op, "fetch", "fetch_size", kernel)
    write = counter_kb(op, "write", "write_size", kernel)
    total[op] = int(2 * fetch + write)   # gfx950: FETCH_SIZE counts half of 16-byte-per-lane reads
    print(op, "HBM bytes per launch", total[op])
    shutil.copy("%s/%s_bench_line.json" % (base, op), "%s/%s_bench_line.json" % (dest, op))
    for line in open("%s/%s_bench_line.json" % (dest, op)):
        if line.startswith("{"):
            d = json.loads(line)
            print(op, "bench %.1f GiB/s, kernel %.2f ms" % (d["value"], d["roofline"]["kernel_ms"]))
j = json.load(open("bench_traffic.json"))
j["decompress:4096x1024KiB:gzip:6"] = total["decompress"]
j["compress:4096x1024KiB:mixed:6"] = total["compress"]
with open("bench_traffic.json", "w") as f:
    json.dump(j, f, indent=1)
    f.write("\n")
for tag in ("sq_inflate", "sq_deflate"):
    fs = glob.glob("gpurun_out/pmc/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        continue
    f = sorted(fs, key=os.path.getmtime)[-1]
    shutil.copy(f, "%s/%s_counters.csv" % (dest, tag))
    acc, n = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "inflate_kernel" in r["Kernel_Name"] or "deflate_chunks" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    print(tag, {k: "%.4g" % (v / n[k]) for k, v in acc.items()})
