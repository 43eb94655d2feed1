#!/usr/bin/env python3
"""A/B timing of goenv_generate between two BUILDS of the library (elf_amd/libgoenv.so against a probe build
elf_amd/libgoenv_ab_<name>.so): child processes alternate between the two, a few rounds each, and every child
also prints a checksum of its last ring so that the two builds are seen to write the same bytes.
Usage: tools/ab_lib_generate.py <name> [boards [plies_per_launch [rounds]]]"""
import os
import subprocess
import sys

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
    for _ in range(4):
        env.generate(T, feat, legal, moves)
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(12):
        env.generate(T, feat, legal, moves)
    env.timer_end()
    us = env.timer_ms() * 1e3 / (12 * T)
    cs = int(feat.view(torch.int32).sum(dtype=torch.int64).item()) ^ int(legal.sum(dtype=torch.int64).item())
    print("%.4f %d" % (us, cs), flush=True)
    sys.exit(0)

name = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 5
libs = {"product": os.path.join(ROOT, "elf_amd", "libgoenv.so"),
        name: os.path.join(ROOT, "elf_amd", "libgoenv_ab_%s.so" % name)}
res = {k: [] for k in libs}
sums = {k: set() for k in libs}
for _ in range(rounds):
    for k, path in libs.items():
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", path, str(B), str(T)],
                           capture_output=True, text=True, timeout=300)
        us, cs = r.stdout.split()[-2:]
        res[k].append(float(us))
        sums[k].add(cs)
for k in libs:
    x = sorted(res[k])
    print("%6d boards x %d plies  %-12s median %.3f us per ply  (%s)  checksum %s" % (
        B, T, k, x[len(x) // 2], " ".join("%.2f" % t for t in res[k]), ",".join(sorted(sums[k]))))
print("same bytes:", sums["product"] == sums[name])
