#!/usr/bin/env python3
"""Turn the output of tools/profile_round.sh into profiles/<round>_pmc.{json,md},
<round>_kernel_stats.csv and <round>_kernel_summary.md.

HBM bytes are corrected as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: FETCH_SIZE and
WRITE_SIZE are reported in KB; FETCH_SIZE is doubled on gfx950.
Usage: tools/summarize_pmc.py [--last=N] gpurun_out/<tag> <round> [kernel boards plies per_ply_bytes per_launch_bytes]
       (--last=N: only the last N launches of the kernel, for commands whose set-up uses the same kernel)
       e.g. ... gpurun_out/r3p r03                                    (the bench line: k_generate_pipe, 1,024 x 128)
            ... gpurun_out/r02g8 r02_generate_8192 "k_generate<3>" 8192 20
            ... gpurun_out/r02p8 r02_ply_8192 "k_ply<3, 1>" 8192 1 36461 3264   (launch per ply)"""
import csv
import json
import os
import shutil
import subprocess
import sys
from collections import defaultdict

KERNEL = "k_generate_pipe"
BOARDS = 1024
PLIES = 128                                  # plies per launch of the profiled command (bench.py --ring)
ALGO_PER_BOARD_PLY = 36100 + 361 + 4         # planes + mask + move, per board per ply
ALGO_PER_BOARD_LAUNCH = 1632 + 1632          # record read + record write, once per launch


LAST = 0                                     # > 0: only the last LAST launches of KERNEL (the timed ones)


def per_launch(path):
    """counter name -> average over the launches of KERNEL (summed over XCDs/dimensions)."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
    ids = sorted({r[0] for r in rows})
    keep = set(ids[-LAST:] if LAST else ids)
    tot, calls = defaultdict(float), defaultdict(set)
    for d, name, v in rows:
        if d in keep:
            tot[name] += v
            calls[name].add(d)
    return {k: tot[k] / len(calls[k]) for k in tot}, max((len(v) for v in calls.values()), default=0)


def main():
    global KERNEL, BOARDS, PLIES, ALGO_PER_BOARD_PLY, ALGO_PER_BOARD_LAUNCH, LAST
    for a in list(sys.argv):
        if a.startswith("--last="):
            LAST = int(a.split("=")[1])
            sys.argv.remove(a)
    src, rnd = sys.argv[1], sys.argv[2]
    if len(sys.argv) > 5:
        KERNEL, BOARDS, PLIES = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    if len(sys.argv) > 7:
        ALGO_PER_BOARD_PLY, ALGO_PER_BOARD_LAUNCH = int(sys.argv[6]), int(sys.argv[7])
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(here, "profiles")
    shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(prof, rnd + "_kernel_stats.csv"))
    subprocess.check_call([sys.executable, os.path.join(here, "tools", "summarize_profile.py"),
                           os.path.join(src, "trace", "t_kernel_trace.csv"),
                           os.path.join(prof, rnd + "_kernel_summary.md")], stdout=subprocess.DEVNULL)
    dur = []
    with open(os.path.join(src, "trace", "t_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                dur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    dur = [d for _, d in sorted(dur)]
    if LAST:
        dur = dur[-LAST:]
    avg_us = sum(dur) / len(dur) / 1e3
    fetch, n = per_launch(os.path.join(src, "fetch", "f_counter_collection.csv"))
    write, _ = per_launch(os.path.join(src, "write", "w_counter_collection.csv"))
    sq, _ = per_launch(os.path.join(src, "sq", "s_counter_collection.csv"))
    wb = write["WRITE_SIZE"] * 1024
    fb = fetch["FETCH_SIZE"] * 1024 * 2
    algo = BOARDS * (PLIES * ALGO_PER_BOARD_PLY + ALGO_PER_BOARD_LAUNCH)
    out = {"kernel": KERNEL.replace(" ", ""), "boards": BOARDS, "plies_per_launch": PLIES, "launches": n,
           "avg_kernel_us": avg_us, "avg_us_per_ply": avg_us / PLIES,
           "write_bytes": wb, "fetch_bytes_corrected": fb, "traffic_bytes": wb + fb,
           "algorithmic_bytes": algo, "ratio": (wb + fb) / algo, "sq": dict(sorted(sq.items()))}
    json.dump(out, open(os.path.join(prof, rnd + "_pmc.json"), "w"), indent=1)
    waves = sq["SQ_WAVES"]
    wc = sq["SQ_WAVE_CYCLES"]
    md = []
    md.append("# HBM traffic and wave-state counters of `%s`, %d boards x %d plies per launch" % (out["kernel"], BOARDS, PLIES))
    md.append("(rocprofv3 --pmc, one pass per counter group, tools/profile_round.sh + tools/summarize_pmc.py; %d launches averaged)\n" % n)
    md.append("| counter | avg per launch | bytes |\n|---|---|---|")
    md.append("| WRITE_SIZE | %.1f KB | %.2f MB (dwordx4 streaming stores: exact per the guide) |" % (write["WRITE_SIZE"], wb / 1e6))
    md.append("| FETCH_SIZE | %.1f KB raw | x2 gfx950 correction = %.2f MB (record loads, history table; narrow widths are uncalibrated: upper bound) |\n" % (fetch["FETCH_SIZE"], fb / 1e6))
    md.append("Algorithmic bytes per launch = %d boards x (%d plies x %d (planes 36,100 + mask 361 [+ move 4]) + %d (record read + record write))" % (BOARDS, PLIES, ALGO_PER_BOARD_PLY, ALGO_PER_BOARD_LAUNCH))
    md.append("= %.2f MB.  Measured traffic = %.2f MB -> %.3fx algorithmic: no wasted re-reads, the kernel is a" % (algo / 1e6, (wb + fb) / 1e6, out["ratio"]))
    md.append("write stream.  Average kernel duration in the same trace: %.2f us = %.2f us per ply.\n" % (avg_us, avg_us / PLIES))
    md.append("Wave-state counters (SQ, summed over the %d waves of a launch):\n" % waves)
    md.append("| counter | per launch | per wave | share of wave cycles |\n|---|---|---|---|")
    for k, label in (("SQ_WAVE_CYCLES", ""), ("SQ_ACTIVE_INST_ANY", " (issuing)"), ("SQ_WAIT_INST_ANY", " (issue stalls)"),
                     ("SQ_WAIT_ANY", " (s_waitcnt / barrier)"), ("SQ_LDS_IDX_ACTIVE", " (LDS-array cycles)"),
                     ("SQ_LDS_BANK_CONFLICT", "")):
        share = "%.0f %%" % (100 * sq[k] / wc) if k.startswith("SQ_W") or k == "SQ_ACTIVE_INST_ANY" else "--"
        if k == "SQ_LDS_BANK_CONFLICT":
            share = "%.0f %% of LDS cycles" % (100 * sq[k] / max(1.0, sq["SQ_LDS_IDX_ACTIVE"]))
        md.append("| %s%s | %.0f | %.0f | %s |" % (k, label, sq[k], sq[k] / waves, share))
    open(os.path.join(prof, rnd + "_pmc.md"), "w").write("\n".join(md) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
