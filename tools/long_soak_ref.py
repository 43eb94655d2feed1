#!/usr/bin/env python3
"""Long soak against the REFERENCE ITSELF (not part of the test suite; needs oracle/_ref/libgoref.so, which travels to
the GPU box): every board of every launch replayed through the unmodified board.cc / board_feature.cc
(goref_batch_check) -- canonical view where the path exposes it, legal mask and all 9,025 floats after every ply --
for millions of board-plies, through the three paths that write planes:
  k_ply<.,1>  launch per ply, on-device policy, games restarted in place     (B boards x PLIES)
  k_ply<.,2>  the fused selfplay step (move sampled from a reply pi)         (B boards x PLIES / 2)
  k_generate_pipe        the bench line's kernel, rings of T plies           (1,024 boards x 64 plies x LAUNCHES)
  k_generate_pipe_multi  several boards per workgroup, D4 codes 0..7         (4,096 boards x 16 plies x LAUNCHES / 2)
Usage: tools/long_soak_ref.py [boards] [plies] [generator launches] [seed]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402
from oracle.pyoracle import RefBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
PLIES = int(sys.argv[2]) if len(sys.argv) > 2 else 800
LAUNCHES = int(sys.argv[3]) if len(sys.argv) > 3 else 40
seed, cap = (int(sys.argv[4]) if len(sys.argv) > 4 else 91), 722
t0 = time.time()
total = 0


def host_state(env):
    return {k: v.cpu().numpy() for k, v in env.get_state().items()}


# --- launch per ply, on-device policy, restarts in place -------------------------------------------------
env = goenv.GoEnv(B, seed=seed, max_ply=cap)
ref = RefBatch(B)
aug = (np.arange(B) % 8).astype(np.int32)
augs = torch.from_numpy(aug).to(env.device)
feat, legal, acts = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361), env.new(torch.int32)
games = np.zeros(B, np.int32)
flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
for t in range(PLIES):
    for b in np.flatnonzero(ref.ended(cap)):              # the device restarts these before this ply
        ref.board(int(b)).clear()
        games[b] += 1
    env.ply(flags, aug_codes=augs, feat_out=feat, legal_out=legal, actions_out=acts)
    a = acts.cpu().numpy()
    assert np.array_equal(a, ref.policy(seed, games)), "ply %d: policy moves differ" % t
    ref.check(a, None, host_state(env) if t % 8 == 0 else None, legal.cpu().numpy(), feat.cpu().numpy(), aug, "k_ply ply %d" % t)
    total += B
    if t % 100 == 0:
        print("k_ply<.,1>: ply %d / %d, %d games restarted, %.0f s" % (t, PLIES, int(games.sum()), time.time() - t0), flush=True)
print("k_ply<.,1>: %d boards x %d plies = %d board-plies, %d games finished: 0 mismatches" % (B, PLIES, B * PLIES, int(games.sum())), flush=True)
env.close()

# --- fused selfplay step (sampled moves are this build's own: the reference replays what the device reports) --
env = goenv.GoEnv(B, seed=seed + 1, max_ply=cap)
ref = RefBatch(B)
feat, legal, acts, status = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361), env.new(torch.int32), env.new(torch.int32)
zero = np.zeros(B, np.int32)
g = torch.Generator(device=env.device)
g.manual_seed(7 + seed)
E = goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END
restarts = 0
for t in range(PLIES // 2):
    pi = torch.rand(B, 361, device=env.device, generator=g)
    env.ply_policy_sel(pi, flags=E, feat_out=feat, legal_out=legal, actions_out=acts, status_out=status)
    a, st = acts.cpu().numpy(), status.cpu().numpy()
    assert (st >= 1).all()
    ref.check(a, None, None, None, None, None, "fused step %d (moves)" % t)   # play the reported moves
    for b in np.flatnonzero(st == 2):                     # the move ended the game: scored, counted, restarted in place
        ref.board(int(b)).clear()
        restarts += 1
    ref.check(None, None, host_state(env) if t % 8 == 0 else None, legal.cpu().numpy(), feat.cpu().numpy(), zero, "fused step %d" % t)
    total += B
print("k_ply<.,2>: %d boards x %d steps = %d board-plies, %d games finished: 0 mismatches" % (B, PLIES // 2, B * (PLIES // 2), restarts), flush=True)
env.close()

# --- the generator: at the bench shape, and with several boards per workgroup under D4 codes ---------------
for GB, T, launches, gcap, with_aug in ((1024, 64, LAUNCHES, cap, False), (4096, 16, max(1, LAUNCHES // 2), 200, True)):
    env = goenv.GoEnv(GB, seed=seed + 2, max_ply=gcap)
    ref = RefBatch(GB)
    fr = torch.empty(T, GB, 25, 19, 19, device=env.device)
    lr = torch.empty(T, GB, 361, dtype=torch.uint8, device=env.device)
    ar = torch.empty(T, GB, dtype=torch.int32, device=env.device)
    games = np.zeros(GB, np.int32)
    gaug = (np.arange(GB) % 8).astype(np.int32) if with_aug else np.zeros(GB, np.int32)
    gaugs = torch.from_numpy(gaug).to(env.device) if with_aug else None
    for L in range(launches):
        fr.fill_(-3.0)
        env.generate(T, fr, lr, ar, aug_codes=gaugs)
        mv = ar.cpu().numpy()
        for t in range(T):
            for b in np.flatnonzero(ref.ended(gcap)):
                ref.board(int(b)).clear()
                games[b] += 1
            assert np.array_equal(mv[t], ref.policy(seed + 2, games)), "launch %d slot %d: moves" % (L, t)
            ref.check(mv[t], None, None, lr[t].cpu().numpy(), fr[t].cpu().numpy(), gaug, "launch %d slot %d" % (L, t))
        total += GB * T
    ref.check(None, None, host_state(env), None, None, None, "records after the last launch")
    print("%s: %d boards x %d plies x %d launches = %d board-plies, %d games finished: 0 mismatches" % (
        env.generate_kernel(), GB, T, launches, GB * T * launches, int(games.sum())), flush=True)
    env.close()
    del fr, lr, ar
print("TOTAL %d board-plies against the reference itself, every float: 0 mismatches, %.0f s" % (total, time.time() - t0), flush=True)
