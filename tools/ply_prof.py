#!/usr/bin/env python3
"""Small workload for rocprofv3 passes over the launch-per-ply kernel: n launches of goenv_ply
(on-device policy, planes + mask) on B mid-game boards, enqueued from C.
Usage: tools/ply_prof.py [boards] [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
env = goenv.GoEnv(B, seed=1, max_ply=722)
feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
env.generate(150, None, None, None)          # into the middle game, by another kernel
torch.cuda.synchronize()
env.ply_repeat(N, R, feat_out=feat, legal_out=legal)
torch.cuda.synchronize()
env.close()
