#!/usr/bin/env python3
"""Diagnostic (never timed as a result): where one fused random ply spends its wave cycles.
Needs elf_amd/libgoenv_diag.so = goenv.hip built with -DGOENV_STAMPS.  Prints the median over
boards of each phase's share.  Usage: tools/stamps.py [boards] [warm plies]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 150
lib = goenv.load_library(os.path.join(ROOT, "elf_amd", "libgoenv_diag.so"))
lib.goenv_diag_ply.restype = C.c_int
lib.goenv_diag_ply.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
lib.goenv_diag_ply_policy.restype = C.c_int
lib.goenv_diag_ply_policy.argtypes = [C.c_void_p, C.c_uint32] + [C.c_void_p] * 5
POLICY = os.environ.get("STAMPS_POLICY") == "1"     # the fused selfplay step (move sampled from pi) instead
env = goenv.GoEnv(B, seed=1)
feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
stamps = torch.zeros((B, 32), dtype=torch.int64, device=env.device)   # [board][wave 0 / 1][16]
pi = torch.ones(B, 361, device=env.device)
for _ in range(warm):
    env.ply(goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART, feat_out=feat, legal_out=legal)
# interval i = stamp i-1 -> stamp i of k_ply (elf_amd/csrc/goenv.hip, STAMP(0)..STAMP(7)): 0 kernel entry,
# 1 record in LDS/registers (+ history preload), 2 move chosen, 3 move applied + liberties rebuilt, 4 end-of-game
# checks done, 5 write_features returned (planes issued), 6 masks of the new position, 7 record + legal mask stored
names = {1: "record load", 2: "move choice", 3: "move + liberties", 4: "game-end checks", 5: "planes",
         6: "masks", 7: "record + legal"}
acc = None
for rep in range(20):
    if POLICY:
        lib.goenv_diag_ply_policy(env._h, goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END, pi.data_ptr(), feat.data_ptr(),
                                  legal.data_ptr(), stamps.data_ptr(), None)
    else:
        lib.goenv_diag_ply(env._h, goenv.F_AUTO_RESTART, feat.data_ptr(), legal.data_ptr(), stamps.data_ptr(), None)
    torch.cuda.synchronize()
    s = stamps.cpu()
    idx = [0, 1, 2, 3, 4, 5, 6, 7]
    d = torch.stack([s[:, idx[i + 1]] - s[:, idx[i]] for i in range(len(idx) - 1)], 1).double()
    acc = d if acc is None else acc + d
acc /= 20
med = acc.median(0).values
tot = (s[:, 7] - s[:, 0]).double()
print("policy-sampled step" if POLICY else "on-device random policy", end=": ")
print("boards", B, "waves/board", os.environ.get("GOENV_WAVES_PER_BOARD", "auto"), "median cycles of wave 0 per phase:")
for i, k in enumerate([1, 2, 3, 4, 5, 6, 7]):
    print("  %-14s %9.0f  (%.1f%%)" % (names[k], med[i].item(), 100 * med[i].item() / med.sum().item()))
print("  total median %.0f  max %.0f" % (tot.median().item(), tot.max().item()))
# (no grid-wide spans: s_memtime is a per-XCD counter, stamps of different workgroups are not comparable)
# inside write_features (slots 13, 14, 15, 11), both recorded waves, cycles after stamp 4 of wave 0
for w in (0, 1):
    base = s[:, 4].double()
    row = ["%s %6.0f" % (nm, (s[:, 16 * w + k].double() - base).median().item())
           for nm, k in (("enter", 4), ("descriptors", 13), ("barrier 1", 14), ("distance / planes issued", 15),
                         ("barrier 2", 11), ("planes 14/15 issued", 5), ("masks", 6), ("record + legal", 7))]
    print("  wave %d, cycles after wave 0 entered write_features: %s" % (w, " | ".join(row)))
