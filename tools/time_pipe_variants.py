#!/usr/bin/env python3
"""Time goenv_generate through k_generate_pipe for several builds of the library (probe builds
elf_amd/libgoenv_ab_<name>.so, -DGOENV_PIPE_ABLATE=n), one child process per build and round.
Usage: tools/time_pipe_variants.py boards plies name[:pipe[:pace_ns]] ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--one":
    import torch
    from elf_amd import goenv
    goenv.load_library(sys.argv[2])
    B, T = int(sys.argv[3]), int(sys.argv[4])
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
    legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
    moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
    env.ply_repeat(300, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    for _ in range(4):
        env.generate(T, feat, legal, moves)
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(12):
        env.generate(T, feat, legal, moves)
    env.timer_end()
    print("%.4f" % (env.timer_ms() * 1e3 / (12 * T)), flush=True)
    sys.exit(0)
B, T = int(sys.argv[1]), int(sys.argv[2])
for spec in sys.argv[3:]:
    name, _, pipe = spec.partition(":")
    pipe, _, pace = pipe.partition(":")
    path = os.path.join(ROOT, "elf_amd", "libgoenv.so" if name == "product" else "libgoenv_ab_%s.so" % name)
    env = dict(os.environ, GOENV_GEN_PIPE=pipe or "0", GOENV_GEN_PACE_NS=pace or "0")
    ts = []
    for _ in range(3):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", path, str(B), str(T)],
                           capture_output=True, text=True, timeout=300, env=env)
        ts.append(float(r.stdout.split()[-1]))
    print("%6d boards x %d plies  %-10s pipe=%s pace=%s  %s us per ply" % (B, T, name, pipe or "0", pace or "0", " ".join("%.2f" % t for t in ts)), flush=True)
