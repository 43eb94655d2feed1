#!/usr/bin/env python3
"""Times the fused ply (and the standalone feature kernel) for every launch shape
(waves per board) over a range of batch sizes; prints a table.  Used to set pick_nw()."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

ALG = 25 * 361 * 4 + 361 + 2 * 1632
sizes = [int(a) for a in sys.argv[1:]] or [256, 1024, 2048, 4096, 8192, 16384, 65536]
print("%8s %3s %10s %10s %9s %10s" % ("boards", "nw", "ply us", "Mmoves/s", "GB/s", "feat us"))
for B in sizes:
    for nw in (1, 2, 3, 6):
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        env.set_waves_per_board(nw)
        feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
        flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
        for _ in range(150):
            env.ply(flags, feat_out=feat, legal_out=legal)
        torch.cuda.synchronize()
        steps = 400 if B <= 8192 else 60
        env.timer_begin()
        for _ in range(steps):
            env.ply(flags, feat_out=feat, legal_out=legal)
        env.timer_end()
        us = env.timer_ms() / steps * 1e3
        env.timer_begin()
        for _ in range(steps):
            env.features(feat, None, legal)
        env.timer_end()
        fus = env.timer_ms() / steps * 1e3
        print("%8d %3d %10.2f %10.1f %9.0f %10.2f" % (B, nw, us, B / us, B * ALG / us * 1e-3, fus), flush=True)
        env.close()
        del feat, legal
