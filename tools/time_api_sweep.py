#!/usr/bin/env python3
"""Fused selfplay step (goenv_ply_policy_sel, one launch) against sampler + ply as two launches
(goenv_sample_policy, goenv_ply_sel), per batch size.  Usage: tools/time_api_sweep.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402

E = goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END
R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART


def timed(env, fn, n, warm=30):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(n):
        fn()
    env.timer_end()
    return env.timer_ms() / n * 1e3


for B in (256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 65536):
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    pi = torch.ones(B, 361, device=env.device)
    acts = torch.zeros(B, dtype=torch.int64, device=env.device)
    env.ply_repeat(100, R)
    n = 400 if B <= 8192 else 60

    def two():
        env.sample_policy(pi, acts)
        env.ply_sel(flags=E, actions=acts, feat_out=feat, legal_out=legal)

    t_r = timed(env, lambda: env.ply(R, feat_out=feat, legal_out=legal), n)
    t_f = timed(env, lambda: env.ply_policy_sel(pi, flags=E, feat_out=feat, legal_out=legal), n)
    t_2 = timed(env, two, n)
    print("%6d boards: random-policy ply %8.2f us | fused step %8.2f us | sampler + ply (2 launches) %8.2f us" % (B, t_r, t_f, t_2),
          flush=True)
    env.close()
    del feat, legal
