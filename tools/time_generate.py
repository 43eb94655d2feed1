#!/usr/bin/env python3
"""Times goenv_generate (persistent multi-ply launch writing planes + masks per ply into ring
buffers) against the launch-per-ply path, for every launch shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ALG = 25 * 361 * 4 + 361
print("%6s %3s %12s %10s %9s" % ("boards", "nw", "us per ply", "Mmoves/s", "GB/s"))
for nw in (1, 2, 3, 6):
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    env.set_waves_per_board(nw)
    feat = torch.zeros(T, B, 25, 19, 19, device=env.device)
    legal = torch.zeros(T, B, 361, dtype=torch.uint8, device=env.device)
    for _ in range(3):
        env.generate(T, feat, legal)
    torch.cuda.synchronize()
    reps = 6
    env.timer_begin()
    for _ in range(reps):
        env.generate(T, feat, legal)
    env.timer_end()
    us = env.timer_ms() * 1e3 / (reps * T)
    print("%6d %3d %12.2f %10.1f %9.0f" % (B, nw, us, B / us, B * ALG / us * 1e-3), flush=True)
    env.close()
    del feat, legal
