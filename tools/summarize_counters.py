#!/usr/bin/env python3
"""Generic per-kernel table of rocprofv3 --pmc passes: every counter averaged per launch, grouped by
(kernel, grid, workgroup) in order of first appearance; launches of one configuration that differ only by a
runtime argument (the probes' variants) are listed one by one with --each.
Usage: tools/summarize_counters.py [--each] <counter_collection.csv> ..."""
import csv
import sys
from collections import defaultdict

each = "--each" in sys.argv
for path in [a for a in sys.argv[1:] if a != "--each"]:
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    order = []
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"], r["Workgroup_Size"])
            if each:
                key = key + (int(r["Dispatch_Id"]),)
            if key not in tot:
                order.append(key)
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[key].add(r["Dispatch_Id"])
    print("##", path)
    for key in order:
        n = len(calls[key])
        if max(tot[key].values()) / n < 1e4:
            continue
        print("%-40s grid %9s wg %4s%s x%d" % (key[0], key[1], key[2], " #%d" % key[3] if each else "", n))
        print("    " + "  ".join("%s %.4g" % (k.replace("_sum", ""), v / n) for k, v in sorted(tot[key].items())))
