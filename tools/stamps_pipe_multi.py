#!/usr/bin/env python3
"""Diagnostic: where the two waves of k_generate_pipe_multi spend their time (100-MHz wall clock, ns per board-turn, mean over the
workgroups).  Needs elf_amd/libgoenv_diag.so (-DGOENV_STAMPS).
Usage: GOENV_GEN_MULTI_M=8 GOENV_GEN_PACE_NS=53000 tools/stamps_pipe_multi.py [boards] [plies per launch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
M = int(os.environ["GOENV_GEN_MULTI_M"])
G = (((B + M - 1) // M + 7) // 8) * 8
lib = goenv.load_library(os.path.join(ROOT, "elf_amd", "libgoenv_diag.so"))
lib.goenv_diag_generate.restype = C.c_int
lib.goenv_diag_generate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
env = goenv.GoEnv(B, seed=1, max_ply=722)
env.ply_repeat(300, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
feat = torch.empty(T, B, 25, 19, 19, device=env.device)
legal = torch.empty(T, B, 361, dtype=torch.uint8, device=env.device)
stamps = torch.zeros((G, 32), dtype=torch.int64, device=env.device)
acc = None
reps = 6
for rep in range(reps + 2):
    lib.goenv_diag_generate(env._h, T, feat.data_ptr(), legal.data_ptr(), stamps.data_ptr(), None)
    torch.cuda.synchronize()
    if rep >= 2:
        s = stamps.cpu().view(G, 2, 16).double()
        acc = s if acc is None else acc + s
acc = acc / reps * 10.0 / (T * M)          # ns per board-turn
names = (("loop / set-up", "loading the board from LDS", "rules of a ply", "saving the board to LDS", "waiting for a free snapshot slot",
          "writing the snapshot", "record write-back"),
         ("expanding a sample", "waiting for a snapshot", "loading it", "rate limiter", "-", "-", "last sample"))
print("%d boards, %d per workgroup (%d workgroups), pace %s ns per ply: ns per board-turn, mean over workgroups" % (B, M, G, os.environ.get("GOENV_GEN_PACE_NS", "0")))
for w, nm in enumerate(("rules wave", "feature wave")):
    m = acc[:, w].mean(0)
    print("  %s: total %.0f" % (nm, m[:7].sum().item()))
    for i in range(7):
        if names[w][i] != "-":
            print("    %-40s %7.0f" % (names[w][i], m[i].item()))
