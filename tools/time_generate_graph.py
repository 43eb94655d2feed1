#!/usr/bin/env python3
"""Does a HIP graph of K consecutive goenv_generate launches cost less per launch boundary than K plain launches?
(pinned pace, so that no event is recorded in either form).  Usage: GOENV_GEN_PACE_NS=6400 tools/time_generate_graph.py [boards plies K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from elf_amd import goenv

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
env = goenv.GoEnv(B, seed=1, max_ply=722)
feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
env.ply_repeat(300, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
for _ in range(4):
    env.generate(T, feat, legal, moves)
torch.cuda.synchronize()


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(reps):
        fn()
    env.timer_end()
    return env.timer_ms() * 1e3 / (reps * K * T)


def plain():
    for _ in range(K):
        env.generate(T, feat, legal, moves)


g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    plain()
for name, fn in (("plain launches", plain), ("graph replay", g.replay), ("plain launches", plain), ("graph replay", g.replay)):
    print("%d boards x %d plies, %d launches per group, pace %s: %-14s %.3f us per ply" % (B, T, K, os.environ.get("GOENV_GEN_PACE_NS"), name, timed(fn, 6)), flush=True)
env.close()
