#!/usr/bin/env python3
"""Per-launch averages of the counters collected by tools/profile_insts.sh for one kernel.
Usage: tools/summarize_insts.py gpurun_out/<tag> <kernel substring> <boards> <plies per launch>"""
import csv
import glob
import sys
from collections import defaultdict

src, kern, boards, plies = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
for path in sorted(glob.glob(src + "/i*/**/*counter_collection.csv", recursive=True)):
    tot, calls = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for r in csv.DictReader(f):
            if kern in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                calls[r["Counter_Name"]].add(r["Dispatch_Id"])
    for k in sorted(tot):
        v = tot[k] / len(calls[k])
        print("%-24s %16.0f per launch  %10.1f per board-ply" % (k, v, v / (boards * plies)))
