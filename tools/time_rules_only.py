#!/usr/bin/env python3
"""Times goenv_generate WITHOUT a plane ring (rules + legal mask + move only) per launch shape: the compute
chain of one board-ply that a storing wave decoupled from the rules would have to hide behind.
Usage: tools/time_rules_only.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B, T = 1024, 64
legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
for nw in (1, 2, 3, 6):
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    env.set_waves_per_board(nw)
    for _ in range(3):
        env.generate(T, None, legal, moves)
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(6):
        env.generate(T, None, legal, moves)
    env.timer_end()
    print("rules + mask + move only, %d boards, nw=%d: %.2f us per ply" % (B, nw, env.timer_ms() * 1e3 / (6 * T)), flush=True)
    env.close()
