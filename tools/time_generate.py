#!/usr/bin/env python3
"""Times goenv_generate (persistent multi-ply launch writing planes + masks per ply into ring
buffers) for every launch shape (waves per board) and both plane-store policies
(GOENV_ZERO_FIRST: 0 = plain stores, 2 = write-through; TG_STORES=0 times the plain ones only).
GOENV_XCD_MAP=0 in the environment turns the XCD-contiguous dealing of rows off.
Usage: tools/time_generate.py [boards [plies_per_launch]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ALG = 25 * 361 * 4 + 361 + 4
print("%6s %3s %5s %12s %10s %9s" % ("boards", "nw", "store", "us per ply", "Mmoves/s", "GB/s"))
feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
for zf in os.environ.get("TG_STORES", "0,2").split(","):
    os.environ["GOENV_ZERO_FIRST"] = zf          # read by goenv_create
    for nw in (1, 2, 3, 6):
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        env.set_waves_per_board(nw)
        for _ in range(3):
            env.generate(T, feat, legal, moves)
        torch.cuda.synchronize()
        reps = 6
        env.timer_begin()
        for _ in range(reps):
            env.generate(T, feat, legal, moves)
        env.timer_end()
        us = env.timer_ms() * 1e3 / (reps * T)
        print("%6d %3d %5s %12.2f %10.1f %9.0f" % (B, nw, "wt" if zf == "2" else "plain", us, B / us, B * ALG / us * 1e-3),
              flush=True)
        env.close()
