#!/usr/bin/env python3
"""Diagnostic: where the two waves of k_generate_pipe spend their time (100-MHz wall clock, summed over a launch,
mean over boards).  Needs elf_amd/libgoenv_diag.so (goenv.hip built with -DGOENV_STAMPS).
Usage: GOENV_GEN_PACE_NS=6200 tools/stamps_pipe.py [boards] [plies per launch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["GOENV_GEN_PIPE"] = "1"
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = goenv.load_library(os.path.join(ROOT, "elf_amd", "libgoenv_diag.so"))
lib.goenv_diag_generate.restype = C.c_int
lib.goenv_diag_generate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
env = goenv.GoEnv(B, seed=1, max_ply=722)
env.ply_repeat(300, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
feat = torch.empty(T, B, 25, 19, 19, device=env.device)
legal = torch.empty(T, B, 361, dtype=torch.uint8, device=env.device)
stamps = torch.zeros((B, 32), dtype=torch.int64, device=env.device)
acc = None
reps = 8
for rep in range(reps + 2):
    lib.goenv_diag_generate(env._h, T, feat.data_ptr(), legal.data_ptr(), stamps.data_ptr(), None)
    torch.cuda.synchronize()
    if rep >= 2:
        s = stamps.cpu().view(B, 2, 16).double()
        acc = s if acc is None else acc + s
acc = acc / reps * 10.0 / T          # ns per ply
names = (("set-up (record load, table copy)", "rules of a ply (move, liberties, masks)", "waiting for a free snapshot slot",
          "writing the snapshot", "record store"),
         ("expanding a sample (plane, mask stores)", "waiting for a snapshot", "loading it", "rate limiter", "last sample"))
print("pace %s ns, %d boards x %d plies: ns per ply, mean over boards" % (os.environ.get("GOENV_GEN_PACE_NS", "0"), B, T))
for w, nm in enumerate(("rules wave", "feature wave")):
    m = acc[:, w].mean(0)
    tot = acc[:, w, :5].sum(1)
    print("  %s: total %.0f  (boards: min %.0f, 10%% %.0f, median %.0f, 90%% %.0f, max %.0f)" % (
        nm, m[:5].sum().item(), tot.min().item(), tot.quantile(0.1).item(), tot.median().item(), tot.quantile(0.9).item(), tot.max().item()))
    for i in range(5):
        print("    %-45s %7.0f" % (names[w][i], m[i].item()))

ftot = acc[:, 1, :5].sum(1)
print("  feature wave total by XCD (workgroup %% 8): %s" % " ".join("%.0f" % ftot[x::8].mean().item() for x in range(8)))
print("  feature wave total by dispatch order within the XCD (workgroup // 8, groups of 16): %s" % " ".join(
    "%.0f" % ftot.view(-1, 8)[i:i + 16].mean().item() for i in range(0, B // 8, 16)))
rt = acc[:, 0, 1]
print("  rules time per ply by board: min %.0f median %.0f max %.0f; correlation of a board's rules time with its feature-wave total: %.2f" % (
    rt.min().item(), rt.median().item(), rt.max().item(), torch.corrcoef(torch.stack([rt, ftot]))[0, 1].item()))
