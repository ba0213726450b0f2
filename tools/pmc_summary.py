#!/usr/bin/env python3
"""Sums the counters of a rocprofv3 --pmc pass per kernel: tools/pmc_summary.py <dir> -> kernel, counter, sum, per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(float)
cnt = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "")
        name = name.split("(")[0].split(" ")[-1][-40:]
        k = (name, r["Counter_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for (kern, c), v in sorted(acc.items()):
    n = len(cnt[(kern, c)])
    print("%-42s %-26s %14.4g per dispatch (%d dispatches)" % (kern, c, v / n, n))
