#!/usr/bin/env python3
"""Small workload for rocprofv3 passes over the STANDALONE feature-extraction kernel (k_features =
BoardFeature::Extract + FindAllValidMoves of B resident mid-game boards into the batch tensors):
n launches of goenv_features.  Prints one JSON line with the HIP-event time per launch.
Usage: tools/features_prof.py [boards] [launches]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
env = goenv.GoEnv(B, seed=1, max_ply=722)
feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
env.generate(150, None, None, None)          # into the middle game, by another kernel
for _ in range(3):
    env.features(feat, None, legal)
torch.cuda.synchronize()
env.timer_begin()
for _ in range(N):
    env.features(feat, None, legal)
env.timer_end()
ms = env.timer_ms() / N
alg = B * (36100 + 361 + 1632)
print(json.dumps({"boards": B, "launches": N, "ms_per_launch": ms, "algorithmic_bytes_per_launch": alg,
                  "GBps": alg / (ms * 1e-3) / 1e9, "frac_of_8TBps": alg / (ms * 1e-3) / 8e12}))
env.close()
