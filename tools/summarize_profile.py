#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): calls, avg/min/max duration.
Usage: tools/summarize_profile.py <kernel_trace.csv> [out.md]"""
import csv
import sys
from collections import defaultdict


def main():
    rows = defaultdict(list)
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"].split("(")[0][:60]
            boards = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
            rows[(name, boards, r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])].append(
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    lines = ["| kernel | workgroups | VGPR | SGPR | LDS B | calls | avg us | min us | max us |", "|---|---|---|---|---|---|---|---|---|"]
    for (name, boards, v, s, l), d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        lines.append("| %s | %d | %s | %s | %s | %d | %.2f | %.2f | %.2f |" % (
            name, boards, v, s, l, len(d), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
