#!/usr/bin/env python3
"""One launch-per-ply step over many boards as ONE launch against several launches over contiguous chunks of the boards, one
after the other on the stream (negative: profiles/r03_ply_chunks_probe.txt; the persistent generator's chunks amortise a launch's front and drain, a one-ply launch cannot).
Usage: tools/time_ply_chunks.py [boards ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elf_amd import goenv  # noqa: E402

R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
for B in [int(a) for a in sys.argv[1:]] or [12288, 16384, 32768, 65536]:
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    env.ply_repeat(150, R)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    row = []
    for chunk in (B, 16384, 8192, 6144, 4096):
        if chunk > B:
            continue
        calls = []
        for first in range(0, B, chunk):
            cnt = min(chunk, B - first)
            calls.append((env._h, None, first, cnt, R, None, None, C.c_void_p(feat[first:].data_ptr()),
                          C.c_void_p(legal[first:].data_ptr()), None, None, None, st))
        fn = env.lib.goenv_ply_sel
        K = 100
        for _ in range(20):
            for a in calls:
                assert fn(*a) == 0
        torch.cuda.synchronize()
        env.timer_begin()
        for _ in range(K):
            for a in calls:
                fn(*a)
        env.timer_end()
        row.append("%d: %7.2f us" % (chunk, env.timer_ms() * 1e3 / K))
    print("%6d boards, launches of <chunk> boards: %s" % (B, " | ".join(row)), flush=True)
    env.close()
    del feat, legal
