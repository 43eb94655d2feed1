#!/usr/bin/env python3
"""A/B timing of goenv_generate inside ONE process: contexts are created alternately under two values of an
environment variable that goenv_create reads (GOENV_XCD_MAP, GOENV_ZERO_FIRST, GOENV_WAVES_PER_BOARD ...), a few
rounds each, so that run-to-run drift of the box (about 2 % between processes) does not drown a small effect.
Usage: tools/ab_generate.py VAR A B [boards [plies_per_launch [rounds]]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

var, a, b = sys.argv[1:4]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
T = int(sys.argv[5]) if len(sys.argv) > 5 else 64
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 5
feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
res = {a: [], b: []}
for _ in range(rounds):
    for v in (a, b):
        os.environ[var] = v
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        for _ in range(4):                       # into the middle game, rings warm
            env.generate(T, feat, legal, moves)
        torch.cuda.synchronize()
        env.timer_begin()
        for _ in range(8):
            env.generate(T, feat, legal, moves)
        env.timer_end()
        res[v].append(env.timer_ms() * 1e3 / (8 * T))
        env.close()
for v in (a, b):
    x = sorted(res[v])
    print("%d boards, %s=%s: median %.3f us per ply  (%s)" % (B, var, v, x[len(x) // 2], " ".join("%.2f" % t for t in res[v])))
