"""Condenses a tools/profile_round.sh run (gpurun_out/prof/) into profiles/<round>/: the kernel statistics as
they are, one small CSV of every counter per kernel (per batch), the bench lines, and bench_traffic.json
(HBM bytes per batch of the engine's own kernels = 2 x FETCH_SIZE + WRITE_SIZE, both in KiB: on gfx950 FETCH_SIZE
counts half of the bytes of 16-byte-per-lane reads).
usage: python tools/collect_profiles.py r02"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
base, dest = "gpurun_out/prof", "profiles/" + rnd
os.makedirs(dest, exist_ok=True)
DOMINANT = {"decompress": ("inflate_decode_kernel", "inflate_expand_kernel", "part_"), "compress": ("deflate_chunks_kernel", "deflate_emit_kernel")}
ONCE = {"decompress": "inflate_decode_kernel", "compress": "deflate_chunks_kernel"}
OURS = ("inflate_", "deflate_", "checksum_", "verify_trailer", "chunk_scan", "seg_scan", "part_")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def newest(pattern):
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1]


rows_out = []
traffic = {}
for op in ("decompress", "compress"):
    f = newest("%s/%s_stats/**/*kernel_stats.csv" % (base, op))
    shutil.copy(f, "%s/%s_kernel_stats.csv" % (dest, op))
    shutil.copy("%s/%s_bench_line.json" % (base, op), "%s/%s_bench_line.json" % (dest, op))
    for r in list(csv.DictReader(open(f)))[:6]:
        print(op, short(r["Name"])[:48], r["Calls"], "%.3f ms" % (float(r["AverageNs"]) / 1e6))
    # per kernel and counter: the sum over its dispatches, the dispatches, and the batches of the pass (a
    # kernel may be dispatched more than once per batch: the second, nearly empty launch of the expand
    # kernel); a batch = one dispatch of the kernel named in ONCE
    per = collections.defaultdict(lambda: [0.0, set(), 1])
    for tag in ("fetch", "write", "sq1", "sq2"):
        # (gpurun merges every run's files into gpurun_out/: only the newest pass of a tag counts)
        for cf in [newest("%s/%s_%s/**/*counter_collection.csv" % (base, op, tag))]:
            rows = list(csv.DictReader(open(cf)))
            batches = len({r["Dispatch_Id"] for r in rows if short(r["Kernel_Name"]).startswith(ONCE[op])}) or 1
            for r in rows:
                k = short(r["Kernel_Name"])
                if not k.startswith(OURS):
                    continue
                e = per[(k, r["Counter_Name"])]
                e[0] += float(r["Counter_Value"])
                e[1].add(r["Dispatch_Id"])
                e[2] = batches
    tot = dom = 0.0
    for (k, c), (v, d, nb) in sorted(per.items()):
        rows_out.append((op, k, c, v / nb, len(d), nb))
        b = 2 * 1024 * v / nb if c == "FETCH_SIZE" else 1024 * v / nb if c == "WRITE_SIZE" else 0.0
        tot += b
        if k.startswith(DOMINANT[op]):
            dom += b
    traffic[op] = (dom, tot)
with open("%s/counters_per_kernel.csv" % dest, "w") as f:
    w = csv.writer(f)
    w.writerow(["bench", "kernel", "counter", "per_batch", "dispatches", "batches"])
    for r in rows_out:
        w.writerow([r[0], r[1], r[2], "%.6g" % r[3], r[4], r[5]])
for tag in ("decompress64k", "compressl12", "compressl1"):   # the other shapes: kernel statistics and bench lines
    found = glob.glob("%s/%s_stats/**/*kernel_stats.csv" % (base, tag), recursive=True)
    if found:
        f = sorted(found, key=os.path.getmtime)[-1]
        shutil.copy(f, "%s/%s_kernel_stats.csv" % (dest, tag))
        shutil.copy("%s/%s_bench_line.json" % (base, tag), "%s/%s_bench_line.json" % (dest, tag))
        for r in list(csv.DictReader(open(f)))[:3]:
            print(tag, short(r["Name"])[:48], r["Calls"], "%.3f ms" % (float(r["AverageNs"]) / 1e6))
sys.path.insert(0, ".")
import bench  # noqa: E402  (the hash of the kernel sources the counters belong to)
j = json.load(open("bench_traffic.json"))
j["csrc_sha16"] = bench.kernel_sources_hash()
j["_comment"] = ("HBM bytes per batch from separate rocprofv3 --pmc passes (profiles/%s/counters_per_kernel.csv): 2 x FETCH_SIZE "
                 "(gfx950 counts half of 16-byte-per-lane reads) + WRITE_SIZE, both in KiB; dominant_kernels = the decoder / match "
                 "and emit kernels, all_kernels = every kernel of the call; valid for the default workloads and for the kernel "
                 "sources whose hash is csrc_sha16" % rnd)
j["decompress:4096x1024KiB:gzip:6"] = {"dominant_kernels": int(traffic["decompress"][0]), "all_kernels": int(traffic["decompress"][1])}
j["compress:4096x1024KiB:mixed:6"] = {"dominant_kernels": int(traffic["compress"][0]), "all_kernels": int(traffic["compress"][1])}
with open("bench_traffic.json", "w") as f:
    json.dump(j, f, indent=1)
    f.write("\n")
print({k: "%.3f GB dominant, %.3f GB all" % (v[0] / 1e9, v[1] / 1e9) for k, v in traffic.items()})
