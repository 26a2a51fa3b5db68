#!/usr/bin/env python3
"""Per-kernel averages of the counters collected by tools/pmc_passes.sh.  Usage: pmc_show.py <tag> <kernel substring>"""
import collections, csv, glob, os, sys
tag, pat = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}", "pass*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void oct::", "").split("(")[0].replace(", ", ",")
        if pat in k:
            a = acc[k][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print(f"    {c:28s} {s / n:16.1f}   (avg of {n} launches)")
