#!/usr/bin/env python3
"""One rules-only ply per launch (goenv_ply without planes; alone, and with the legal mask and the move written) at 1 / 2 / 3
waves per board: the measurement behind pick_nw_rules in goenv.hip.  Usage: tools/time_rules_waves.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elf_amd import goenv  # noqa: E402

R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
for B in (1024, 2048, 4096, 8192, 16384, 65536):
    row = []
    for NW in (1, 2, 3):
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        env.set_waves_per_board(NW)
        legal = env.new(torch.uint8, 361)
        acts = env.new(torch.int32)
        env.ply_repeat(150, R)
        torch.cuda.synchronize()
        K = 300
        env.timer_begin(); env.ply_repeat(K, R); env.timer_end()
        a = env.timer_ms() * 1e3 / K
        env.timer_begin()
        for _ in range(K):
            env.ply(R, legal_out=legal, actions_out=acts)
        env.timer_end()
        b = env.timer_ms() * 1e3 / K
        row.append("NW%d %7.2f / %7.2f us" % (NW, a, b))
        env.close()
    print("%6d boards, rules only / + legal mask and move out: %s" % (B, " | ".join(row)), flush=True)
