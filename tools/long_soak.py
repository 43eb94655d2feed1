#!/usr/bin/env python3
"""Long soak (not part of the test suite): every float of every board on every ply against oracle boards,
through the fused selfplay step (goenv_ply_policy_sel, write-through stores at 1,024 boards) and through
the generator, for many more plies than tests/test_gpu_soak.py.  Usage: tools/long_soak.py [boards] [steps]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402
from oracle.pyoracle import OracleBoard  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed, cap = 77, 300
t0 = time.time()
env = goenv.GoEnv(B, seed=seed, max_ply=cap)
dev = env.device
feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
acts = env.new(torch.int32)
oracles = [OracleBoard() for _ in range(B)]
game = [0] * B
rng = np.random.RandomState(3)
bad = 0
E = goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END
for step in range(STEPS):
    pi = rng.rand(B, 361).astype(np.float32)
    pi[rng.randint(B)] = 0.0
    feat.fill_(-3.0)
    env.ply_policy_sel(torch.from_numpy(pi).to(dev), flags=E, feat_out=feat, legal_out=legal, actions_out=acts)
    f, lg, a = feat.cpu().numpy().reshape(B, -1), legal.cpu().numpy(), acts.cpu().tolist()
    for b in range(B):
        o = oracles[b]
        want = o.sample_policy(seed, b, game[b], pi[b], 0)
        if a[b] != want:
            bad += 1
            print("step %d board %d: move %d, oracle %d" % (step, b, a[b], want), flush=True)
        ok, end = o.forward(want)
        assert ok
        if end or o.canonical()["scalars"][0] > cap:
            o = oracles[b] = OracleBoard()
            game[b] += 1
        if not np.array_equal(f[b].view(np.uint32), o.extract(0).view(np.uint32)) or not np.array_equal(lg[b], o.legal()):
            bad += 1
            print("step %d board %d: planes / mask differ" % (step, b), flush=True)
    if step % 50 == 0:
        print("fused step %d / %d, %d mismatches, %.0f s" % (step, STEPS, bad, time.time() - t0), flush=True)
print("fused selfplay step: %d board-plies, %d mismatches, games finished %d" % (B * STEPS, bad, sum(game)), flush=True)
env.close()

T = 64
env = goenv.GoEnv(B, seed=seed, max_ply=cap)
fr = torch.empty(T, B, 25, 19, 19, device=dev)
lr = torch.empty(T, B, 361, dtype=torch.uint8, device=dev)
ar = torch.empty(T, B, dtype=torch.int32, device=dev)
oracles = [OracleBoard() for _ in range(B)]
game = [0] * B
bad2 = 0
for launch in range(max(1, STEPS // T)):
    fr.fill_(-3.0)
    env.generate(T, fr, lr, ar)
    mv = ar.cpu().numpy()
    for t in range(T):
        f, lg = fr[t].cpu().numpy().reshape(B, -1), lr[t].cpu().numpy()
        for b in range(B):
            o = oracles[b]
            if o.is_game_end() or o.canonical()["scalars"][0] > cap:
                o = oracles[b] = OracleBoard()
                game[b] += 1
            if int(mv[t, b]) != o.policy_move(seed, b, game[b]):
                bad2 += 1
            assert o.forward(int(mv[t, b]))[0]
            if not np.array_equal(f[b].view(np.uint32), o.extract(0).view(np.uint32)) or not np.array_equal(lg[b], o.legal()):
                bad2 += 1
    print("generator launch %d, %d mismatches, %.0f s" % (launch, bad2, time.time() - t0), flush=True)
print("generator: %d board-plies, %d mismatches, games finished %d" % (B * T * max(1, STEPS // T), bad2, sum(game)), flush=True)
sys.exit(1 if bad or bad2 else 0)
