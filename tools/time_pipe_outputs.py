#!/usr/bin/env python3
"""k_generate_pipe at pinned paces with all outputs (planes + legal mask + move) against planes only: do the small,
line-sharing stores of masks and moves move the pace that the memory side sustains?  Child process per point.
Usage: tools/time_pipe_outputs.py [boards [plies per launch]]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from elf_amd import goenv
    B, T, what = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
    legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda") if "l" in what else None
    moves = torch.zeros(T, B, dtype=torch.int32, device="cuda") if "a" in what else None
    env.ply_repeat(100, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    for _ in range(4):
        env.generate(T, feat, legal, moves)
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(12):
        env.generate(T, feat, legal, moves)
    env.timer_end()
    print("%.3f" % (env.timer_ms() * 1e3 / (12 * T)), flush=True)
    sys.exit(0)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
for what in ("fla", "f", "fl", "fa"):
    row = []
    for pace in (5900, 6000, 6100, 6180, 6250, 6400):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B), str(T), what], capture_output=True, text=True,
                           timeout=200, env=dict(os.environ, GOENV_GEN_PACE_NS=str(pace * B // 1024)))
        row.append("%d: %s" % (pace, r.stdout.split()[-1] if r.returncode == 0 else "fail"))
    print("%d boards x %d plies, outputs %-3s, us per ply at pace (ns per ply of 1,024 boards): %s" % (B, T, what, " | ".join(row)), flush=True)
