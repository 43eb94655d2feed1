#!/usr/bin/env python3
"""goenv_score (Tromp-Taylor + fast score of every board) and goenv_get_state at 1 / 2 / 3 / 6 waves per board, mid-game
positions.  Usage: tools/time_score_waves.py [boards ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elf_amd import goenv  # noqa: E402

R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
for B in [int(a) for a in sys.argv[1:]] or [256, 1024, 4096, 16384, 65536]:
    row = []
    for NW in (1, 2, 3, 6):
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        env.ply_repeat(200, R)
        env.set_waves_per_board(NW)
        tt, fast = env.new(torch.float32), env.new(torch.float32)
        for _ in range(5):
            env.score(tt, fast)
        torch.cuda.synchronize()
        K = 50
        env.timer_begin()
        for _ in range(K):
            env.score(tt, fast)
        env.timer_end()
        row.append("NW%d %8.2f us" % (NW, env.timer_ms() * 1e3 / K))
        env.close()
    print("%6d boards, goenv_score: %s" % (B, " | ".join(row)), flush=True)
