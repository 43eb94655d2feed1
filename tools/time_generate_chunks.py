#!/usr/bin/env python3
"""goenv_generate over more boards than k_generate_pipe_multi holds: launches over chunks of the boards (default) against
k_generate<3> over all of them (GOENV_GEN_NO_CHUNKS=1); child processes, a checksum of the last ring from each.
Usage: tools/time_generate_chunks.py [boards[:plies per launch] ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from elf_amd import goenv
    B, T = int(sys.argv[2]), int(sys.argv[3])
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
    legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
    moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
    env.ply_repeat(100, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    for _ in range(24):
        env.generate(T, feat, legal, moves)
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(12):
        env.generate(T, feat, legal, moves)
    env.timer_end()
    us = env.timer_ms() * 1e3 / (12 * T)
    cs = int(feat.view(torch.int32).sum(dtype=torch.int64).item()) ^ int(legal.sum(dtype=torch.int64).item()) ^ int(moves.sum(dtype=torch.int64).item())
    print("%.3f %d %s %.2f" % (us, cs, env.generate_kernel(), env.generate_pace()[0]), flush=True)
    sys.exit(0)

for arg in sys.argv[1:] or ["16384:8", "16384:16", "32768:8", "65536:4", "65536:8", "65536:16"]:   # boards[:plies per launch]
    B, T = (int(x) for x in (arg.split(":") + ["8"])[:2])
    out = {}
    for name, envv in (("chunks", {}), ("k_generate<3>", {"GOENV_GEN_NO_CHUNKS": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B), str(T)], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, **envv))
        out[name] = r.stdout.split()[-4:] if r.returncode == 0 else ["fail", r.stderr[-300:], "", ""]
    print("%6d boards x %d plies per launch: %s" % (B, T, " | ".join("%s %s us per ply (%s, pace %s)" % (k, v[0], v[2], v[3]) for k, v in out.items())),
          " same bytes:", out["chunks"][1] == out["k_generate<3>"][1], flush=True)
