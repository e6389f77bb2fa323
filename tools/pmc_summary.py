#!/usr/bin/env python3
"""Mean per launch of every counter in rocprofv3 --pmc CSV output, per kernel-name substring.
usage: tools/pmc_summary.py <dir with *counter_collection.csv> <kernel substring> [...]"""
import csv
import glob
import sys
from collections import defaultdict

root, wanted = sys.argv[1], sys.argv[2:]
sums = defaultdict(lambda: defaultdict(float))
launches = defaultdict(lambda: defaultdict(set))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        for w in wanted:
            if w in name:
                sums[w][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[w][r["Counter_Name"]].add((path, r["Dispatch_Id"]))
for w in wanted:
    print(w)
    for c in sorted(sums[w]):
        n = len(launches[w][c])
        print(f"  {c:28s} {sums[w][c] / n:14.4g}   ({n} launches)")
