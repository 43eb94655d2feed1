#!/usr/bin/env python3
"""goenv_generate with its default settings (pipelined kernel, self-tuning rate limiter) against GOENV_GEN_PIPE=0:
time per ply over successive groups of launches, and what the tuner settled on.  Usage: tools/time_generate_tuned.py [boards plies]"""
import os, subprocess, sys
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
    env.ply_repeat(300, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    torch.cuda.synchronize()
    out = []
    for group in range(8):
        env.timer_begin()
        for _ in range(12):
            env.generate(T, feat, legal, moves)
        env.timer_end()
        out.append(env.timer_ms() * 1e3 / (12 * T))
    pace, est = env.generate_pace()
    print("%5d boards x %3d plies %-18s us per ply by group of 12 launches: %s | pace %.2f ns/board, estimates %s" % (
        B, T, "GOENV_GEN_PIPE=" + os.environ.get("GOENV_GEN_PIPE", "default"), " ".join("%.2f" % x for x in out), pace,
        " ".join("%.2f" % e for e in est)), flush=True)
    sys.exit(0)
shapes = [(int(sys.argv[1]), int(sys.argv[2]))] if len(sys.argv) > 2 else [(256, 64), (512, 64), (1024, 64), (1536, 32), (2048, 32), (2304, 32), (3072, 16)]
for B, T in shapes:
    for pipe in ("0", None):
        env = dict(os.environ)
        env.pop("GOENV_GEN_PIPE", None)
        if pipe is not None:
            env["GOENV_GEN_PIPE"] = pipe
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B), str(T)], capture_output=True, text=True, timeout=300, env=env)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:], flush=True)
