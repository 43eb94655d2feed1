#!/usr/bin/env python3
"""Diagnostic (never timed as a result): where one ply of the persistent generator spends its wave
cycles, for wave 0 and wave 1 of a board.  Needs elf_amd/libgoenv_diag.so =
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DGOENV_STAMPS -o elf_amd/libgoenv_diag.so elf_amd/csrc/goenv.hip
Prints, per wave, the median over boards of the cycles between consecutive stamps of the LAST ply of
a launch.  Usage: tools/stamps_generate.py [boards] [plies per launch]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = goenv.load_library(os.path.join(ROOT, "elf_amd", "libgoenv_diag.so"))
lib.goenv_diag_generate.restype = C.c_int
lib.goenv_diag_generate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
env = goenv.GoEnv(B, seed=1, max_ply=722)
feat = torch.empty(T, B, 25, 19, 19, device=env.device)
legal = torch.empty(T, B, 361, dtype=torch.uint8, device=env.device)
stamps = torch.zeros((B, 32), dtype=torch.int64, device=env.device)
# order of the stamps within one ply: loop top, move picked, move applied + liberties rebuilt, masks,
# masks gathered, legal written, [features: descriptors scattered, barrier passed, planes 0-11 / distance
# transform done, barrier passed], planes 14/15 issued
order = [8, 2, 3, 9, 10, 11, 13, 14, 15, 7, 12]
names = ["pick move", "apply + rebuild liberties", "move masks", "gather masks", "write legal", "descriptors + ballots",
         "wait at barrier 1", "planes 0-11 + zeros (wave 1) / distance transform (wave 0)", "wait at barrier 2",
         "planes 14 / 15"]
acc = None
reps = 12
for rep in range(reps + 2):
    lib.goenv_diag_generate(env._h, T, feat.data_ptr(), legal.data_ptr(), stamps.data_ptr(), None)
    torch.cuda.synchronize()
    if rep < 2:
        continue
    s = stamps.cpu().view(B, 2, 16)
    d = torch.stack([s[:, :, order[i + 1]] - s[:, :, order[i]] for i in range(len(order) - 1)], 2).double()
    acc = d if acc is None else acc + d
acc /= reps
for w in range(2):
    med = acc[:, w].median(0).values
    print("boards %d, plies per launch %d, wave %d: median cycles per phase of the last ply (s_memtime, 100 MHz ticks x clock ratio as reported)" % (B, T, w))
    for i, nm in enumerate(names):
        print("  %-62s %8.0f  (%4.1f%%)" % (nm, med[i].item(), 100 * med[i].item() / med.sum().item()))
    print("  %-62s %8.0f" % ("sum", med.sum().item()))
