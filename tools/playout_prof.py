#!/usr/bin/env python3
"""Workload for rocprofv3 passes over the RULES-ONLY step kernel (k_playout: whole random games with the
board in LDS/registers, no planes): `launches` launches of goenv_playout on B fresh boards, timed with HIP
events on the launch stream; prints one JSON line (moves, ms, moves/s per launch).
Usage: tools/playout_prof.py [boards] [launches] [waves per board, 0 = the library's choice]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 0
env = goenv.GoEnv(B, seed=1, max_ply=722)
if NW:
    env.set_waves_per_board(NW)
nm = env.new(torch.int32)
out = []
for i in range(N + 1):                       # launch 0 is warm-up (code object load)
    env.reset()
    torch.cuda.synchronize()
    env.timer_begin()
    env.playout(None, nm)
    env.timer_end()
    ms = env.timer_ms()
    moves = int(nm.sum().item())
    if i:
        out.append({"ms": ms, "moves": moves, "moves_per_s": moves / (ms * 1e-3)})
env.close()
avg = sum(o["moves_per_s"] for o in out) / len(out)
print(json.dumps({"boards": B, "waves_per_board": NW, "launches": out, "moves_per_s": avg,
                  "plies_per_board": out[0]["moves"] / B}))
