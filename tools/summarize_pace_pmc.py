#!/usr/bin/env python3
"""Average per launch of k_generate_pipe of every counter collected by tools/profile_pace_pmc.sh, one column per pace.
Usage: tools/summarize_pace_pmc.py gpurun_out/<tag>"""
import csv, glob, os, re, sys
from collections import defaultdict
src = sys.argv[1]
table = defaultdict(dict)
for path in sorted(glob.glob(os.path.join(src, "p*_g*", "*counter_collection.csv"))):
    pace = int(re.search(r"/p(\d+)_g", path).group(1))
    tot, calls = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if "k_generate_pipe" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                calls[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in tot:
        table[k][pace] = tot[k] / len(calls[k])
paces = sorted({p for v in table.values() for p in v})
print("%-40s %s" % ("counter (per launch: 1,024 boards x 64 plies)", " ".join("%16s" % ("pace %d ns" % p if p else "free-running") for p in paces)))
for k in sorted(table):
    print("%-40s %s" % (k, " ".join("%16.0f" % table[k].get(p, float("nan")) for p in paces)))
for p in paces:
    try:
        print("us per ply at %d: %s" % (p, open(os.path.join(src, "p%d_g1.txt" % p)).read().split()[-1]))
    except (OSError, IndexError):
        pass
