#!/usr/bin/env python3
"""Small workload for rocprofv3 passes: the fused selfplay step (k_ply<NW,2>) and the on-device-policy
ply (k_ply<NW,1>) over the same boards, planes on.  Usage: tools/api_step_prof.py [boards] [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
env = goenv.GoEnv(B, seed=1, max_ply=722)
feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
pi = torch.ones(B, 361, device=env.device)
R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
E = goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END
env.ply_repeat(100, R, feat_out=feat, legal_out=legal)
for _ in range(N):
    env.ply(R, feat_out=feat, legal_out=legal)
for _ in range(N):
    env.ply_policy_sel(pi, flags=E, feat_out=feat, legal_out=legal)
acts = torch.zeros(B, dtype=torch.int64, device=env.device)
for _ in range(N):                       # k_sample_policy + k_ply<NW,0>: the sampler outside the ply kernel
    env.sample_policy(pi, acts)
    env.ply_sel(flags=E, actions=acts, feat_out=feat, legal_out=legal)
torch.cuda.synchronize()
env.close()
