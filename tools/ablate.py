#!/usr/bin/env python3
"""Diagnostic: kernel time of the fused ply with parts compiled out (results are WRONG in these
builds; only the timing is read).  Needs elf_amd/libgoenv_ab_*.so."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    from elf_amd import goenv
    goenv.load_library(sys.argv[2])
    B = int(sys.argv[3])
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    env.ply_repeat(150, flags, feat_out=feat, legal_out=legal)
    torch.cuda.synchronize()
    env.timer_begin()
    env.ply_repeat(500, flags, feat_out=feat, legal_out=legal)      # launches enqueued from C: no host gaps
    env.timer_end()
    print("%-28s %6d boards  %.2f us" % (os.path.basename(sys.argv[2]), B, env.timer_ms() * 2), flush=True)
    sys.exit(0)

for B in [int(x) for x in os.environ.get('ABLATE_BOARDS', '1024').split(',')]:
    for name in ["libgoenv.so"] + sorted(f for f in os.listdir(os.path.join(ROOT, "elf_amd")) if f.startswith("libgoenv_ab_")):
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", os.path.join(ROOT, "elf_amd", name), str(B)],
                       timeout=120)
