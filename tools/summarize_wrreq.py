#!/usr/bin/env python3
"""Per kernel and launch configuration: write requests at the L2 -> memory interface, the share that are 64-B
requests, bytes, and request-stall cycles, from the passes of tools/profile_wrreq.sh.
Usage: tools/summarize_wrreq.py gpurun_out/<tag>"""
import csv
import glob
import os
import sys
from collections import defaultdict

src = sys.argv[1]
for path in sorted(glob.glob(os.path.join(src, "*", "*counter_collection.csv"))):
    tot = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    order = []
    with open(path) as f:
        for r in csv.DictReader(f):
            key = (r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"], r["Workgroup_Size"])
            if key not in tot:
                order.append(key)
            tot[key][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[key].add(r["Dispatch_Id"])
    print("##", os.path.relpath(path, src))
    for key in order:
        c, n = tot[key], len(calls[key])
        req = c.get("TCC_EA0_WRREQ_sum", 0) / n
        if req < 1e5:
            continue
        r64 = c.get("TCC_EA0_WRREQ_64B_sum", 0) / n
        mb = ((req - r64) * 32 + r64 * 64) / 1e6
        print("%-62s grid %9s wg %4s launches %3d  requests %11.0f  64-B share %.3f  %8.1f MB  WRREQ_STALL %12.0f  TOO_MANY_EA_WRREQS %12.0f"
              % (key[0], key[1], key[2], n, req, r64 / req, mb, c.get("TCC_EA0_WRREQ_STALL_sum", 0) / n,
                 c.get("TCC_TOO_MANY_EA_WRREQS_STALL_sum", 0) / n))
