#!/usr/bin/env python3
"""Average PMC counter values per launch of the kernels whose name contains a pattern, from a rocprofv3 --pmc run:
    python tools/pmc_kernel.py <output dir> <pattern>"""
import collections
import csv
import glob
import sys

root, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        if pat not in name:
            continue
        key = name[:60]
        acc[key][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(key, row["Counter_Name"])] += 1
for key, counters in acc.items():
    print(key)
    for c, v in sorted(counters.items()):
        print(f"   {c:32s} {v / cnt[(key, c)]:16.1f}  (avg over {cnt[(key, c)]} launches)")
