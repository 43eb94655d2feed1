#!/usr/bin/env python3
"""Times the launch-per-ply kernel and the standalone feature kernel with the XCD-contiguous dealing of
output rows (xcd_row, elf_amd/csrc/goenv.hip) forced off and on, per batch size.
Usage: tools/time_rowmap.py [boards ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192, 16384]
flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
print("%6s %4s %12s %14s" % ("boards", "map", "ply us", "features us"))
for B in sizes:
    feat = torch.zeros(B, 25, 19, 19, device="cuda")
    legal = torch.zeros(B, 361, dtype=torch.uint8, device="cuda")
    for m in ("0", "1"):
        os.environ["GOENV_XCD_MAP"] = m          # read by goenv_create
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        steps = max(50, min(2000, 2000 * 1024 // B))
        env.ply_repeat(100, flags, feat_out=feat, legal_out=legal)
        torch.cuda.synchronize()
        env.timer_begin()
        env.ply_repeat(steps, flags, feat_out=feat, legal_out=legal)
        env.timer_end()
        ply = env.timer_ms() * 1e3 / steps
        steps = min(steps, 300)
        for _ in range(10):
            env.features(feat, None, legal)
        env.timer_begin()
        for _ in range(steps):
            env.features(feat, None, legal)
        env.timer_end()
        print("%6d %4s %12.2f %14.2f" % (B, m, ply, env.timer_ms() * 1e3 / steps), flush=True)
        env.close()
