#!/usr/bin/env python3
"""Turns the passes of tools/profile_playout.sh into profiles/<round>_playout_<boards>.json and one
markdown table (profiles/<round>_playout.md): the rules-only step kernel (k_playout) against gfx950's
peaks (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 8 waves per SIMD, one wave64 VALU instruction per 2
cycles per SIMD, 128 B/clk/CU of LDS).

Units (MI355X_MICROARCH.md, constants table): GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles; / 8 =
the kernel's length in shader cycles, which / its traced duration gives the clock held);
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_LDS_IDX_ACTIVE and
SQ_LDS_BANK_CONFLICT count LDS-array cycles summed over CUs.
Usage: tools/summarize_playout.py <round> gpurun_out/<tag> [gpurun_out/<tag> ...]"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

CUS, SIMDS, XCDS = 256, 1024, 8
KERNEL = "k_playout"


def per_launch(path):
    tot, calls = defaultdict(float), defaultdict(set)
    geo = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"])
                calls[r["Counter_Name"]].add(r["Dispatch_Id"])
                geo = {"kernel": r["Kernel_Name"].split("(")[0].replace("void ", ""), "vgpr": int(r["VGPR_Count"]),
                       "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                       "workgroup": int(r["Workgroup_Size"])}
    return {k: tot[k] / len(calls[k]) for k in tot}, geo


def one(src):
    c, geo = {}, {}
    for sub in ("sq/s", "i1/i", "i2/i"):
        d, g = per_launch(os.path.join(src, sub + "_counter_collection.csv"))
        for k, v in d.items():
            c.setdefault(k, v)
        geo = g or geo
    dur = []
    with open(os.path.join(src, "trace", "t_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            if KERNEL in r["Kernel_Name"]:
                dur.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    dur = [d for _, d in sorted(dur)][1:]                 # launch 0 is the warm-up
    us = sum(dur) / len(dur) / 1e3
    plain = json.load(open(os.path.join(src, "plain.json")))
    boards, moves = plain["boards"], plain["launches"][0]["moves"]
    cyc = c["GRBM_GUI_ACTIVE"] / XCDS
    nw = geo["workgroup"] // 64
    wg_per_cu = min(32 // nw, (160 * 1024) // max(1, geo["lds_bytes"]))
    out = dict(
        boards=boards, kernel=geo["kernel"], waves_per_board=nw, vgpr=geo["vgpr"], sgpr=geo["sgpr"],
        lds_bytes_per_workgroup=geo["lds_bytes"], kernel_us_profiled=us, kernel_ms_unprofiled=plain["launches"][-1]["ms"],
        moves_per_launch=moves, moves_per_s=plain["moves_per_s"], plies_per_board=plain["plies_per_board"],
        kernel_cycles=cyc, clock_ghz=cyc / (us * 1e3),
        waves_launched=c["SQ_WAVES"], resident_wave_limit_per_simd=min(8.0, wg_per_cu * nw / 4.0),
        waves_per_simd=c["SQ_WAVE_CYCLES"] * 4 / (cyc * SIMDS), waves_per_simd_peak=8,
        valu_insts_per_simd_cycle=c["SQ_INSTS_VALU"] / (cyc * SIMDS), valu_issue_peak=0.5,
        valu_busy_gfx94x_formula=c["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * SIMDS),
        salu_insts_per_simd_cycle=c["SQ_INSTS_SALU"] / (cyc * SIMDS),
        lds_busy=c["SQ_LDS_IDX_ACTIVE"] / (cyc * CUS), lds_bank_conflict_share=c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
        wave_state=dict(parked_waitcnt_barrier=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                        issue_stall=c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                        issuing=c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                        lds_issue_stall=c.get("SQ_WAIT_INST_LDS", 0) / c["SQ_WAVE_CYCLES"]),
        per_board_ply=dict(valu=c["SQ_INSTS_VALU"] / moves, salu=c["SQ_INSTS_SALU"] / moves, lds=c["SQ_INSTS_LDS"] / moves,
                           lds_array_cycles=c["SQ_LDS_IDX_ACTIVE"] / moves,
                           # SQ_ACTIVE_INST_VALU == SQ_INSTS_VALU in every pass: one count per instruction, so
                           # SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = lanes enabled per instruction
                           active_lanes_per_valu=c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
                           if c.get("SQ_THREAD_CYCLES_VALU") else None),
        counters={k: c[k] for k in sorted(c)})
    out["valu_issue_frac"] = out["valu_insts_per_simd_cycle"] / 0.5
    return out


def main():
    rnd, srcs = sys.argv[1], sys.argv[2:]
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    rows = []
    for s in srcs:
        o = one(s)
        rows.append(o)
        with open(os.path.join(prof, "%s_playout_%d.json" % (rnd, o["boards"])), "w") as f:
            json.dump(o, f, indent=1)
        shutil.copy(os.path.join(s, "trace", "t_kernel_stats.csv"), os.path.join(prof, "%s_playout_%d_kernel_stats.csv" % (rnd, o["boards"])))
    with open(os.path.join(prof, rnd + "_playout.md"), "w") as f:
        f.write("# Rules-only step kernel (`k_playout`: whole random games, board in LDS/registers, no planes) against gfx950 peaks\n\n"
                "rocprofv3 `--kernel-trace --stats` + three `--pmc` passes (tools/profile_playout.sh, tools/summarize_playout.py); "
                "per launch, averages over the launches after the warm-up.  Peaks: 8 waves per SIMD (1,024 SIMDs), one wave64 "
                "VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md: SIMD-32), LDS array busy every cycle on every CU.\n\n")
        f.write("| boards | kernel | moves/s (unprofiled) | kernel time | clock | waves/SIMD achieved (limit of this launch / peak) | "
                "VALU instr per SIMD-cycle (of peak 0.5) | `VALUBusy` (gfx94x formula) | SALU instr per SIMD-cycle | LDS array busy | "
                "bank-conflict share of LDS cycles | wave cycles: parked / issue-stalled / issuing |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for o in rows:
            w = o["wave_state"]
            f.write("| %d | `%s`, %.1f KB LDS per board | %.0f M | %.2f ms | %.2f GHz | %.2f (%.1f / 8) | %.3f (%.0f %%) | %.0f %% | %.3f | %.0f %% | %.0f %% | %.0f / %.0f / %.0f %% |\n" % (
                o["boards"], o["kernel"], o["lds_bytes_per_workgroup"] / 1024, o["moves_per_s"] / 1e6,
                o["kernel_ms_unprofiled"], o["clock_ghz"], o["waves_per_simd"], o["resident_wave_limit_per_simd"],
                o["valu_insts_per_simd_cycle"], 100 * o["valu_issue_frac"], 100 * o["valu_busy_gfx94x_formula"],
                o["salu_insts_per_simd_cycle"], 100 * o["lds_busy"], 100 * o["lds_bank_conflict_share"],
                100 * w["parked_waitcnt_barrier"], 100 * w["issue_stall"], 100 * w["issuing"]))
        f.write("\nPer board-ply (wave instructions summed over the board's waves): ")
        f.write("; ".join("%d boards: %.0f VALU, %.0f SALU, %.0f LDS instructions, %.0f LDS-array cycles, %.1f of 64 lanes active per VALU instruction"
                          % (o["boards"], o["per_board_ply"]["valu"], o["per_board_ply"]["salu"], o["per_board_ply"]["lds"],
                             o["per_board_ply"]["lds_array_cycles"], o["per_board_ply"]["active_lanes_per_valu"] or 0) for o in rows))
        f.write(".\n")
        if all(o["waves_per_board"] == 3 for o in rows):   # the reading below was written for the three-wave launches of r03
            f.write("\nReading: the kernel is bound by instruction ISSUE and by the latency of one board's dependent chain, not by "
                    "a memory system: at full occupancy (65,536 boards) 37 % of the wave cycles are parked on `s_waitcnt` / barriers, 37 % "
                    "wait for an issue slot and 26 % issue; the vector pipe issues at 35 % of its peak rate (69 % by the gfx94x `VALUBusy` "
                    "formula, which prices an instruction at 4 cycles), the LDS array is busy a third of the time with a tenth of that "
                    "lost to bank conflicts, and there are MORE scalar than vector instructions (the 361-bit masks live in scalar "
                    "registers: ballots, popcounts, 64-bit mixes of the counter RNG).  The guide documents no scalar issue peak; under "
                    "the classic one-issue-turn-per-SIMD-every-fourth-cycle model (an inference, not a measured constant) 0.25 scalar "
                    "instructions per SIMD-cycle would be the ceiling and the kernel sits at 0.19 of it at 65,536 boards.  At 1,024 boards the "
                    "launch is 2.2 waves per SIMD of a possible 7.5 (3,072 waves on 1,024 SIMDs): occupancy, i.e. the batch size, is the limit.\n")
    print(open(os.path.join(prof, rnd + "_playout.md")).read())


if __name__ == "__main__":
    main()
