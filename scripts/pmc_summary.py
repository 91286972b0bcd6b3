#!/usr/bin/env python3
"""Average per launch of every counter in rocprofv3 --pmc output directories, by kernel (library kernels only).
  python scripts/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            if "k_" in n and "at::" not in n:
                agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print("   %-28s %16.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
