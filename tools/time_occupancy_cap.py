#!/usr/bin/env python3
"""Probe: do the large free-running launches (k_ply, k_features at 8,192 .. 65,536 boards) gain from FEWER workgroups per CU
(less offered store traffic)?  GOENV_LDS_PAD=<bytes> adds unused dynamic LDS per workgroup.  One child process per setting.
Usage: tools/time_occupancy_cap.py boards [pad_bytes ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if sys.argv[1] == "--one":
    import torch
    from elf_amd import goenv
    B = int(sys.argv[2])
    N = max(6, min(40, (1 << 20) // B))
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    env.generate(150, None, None, None)
    out = []
    for what in ("ply", "features"):
        def run(n):
            if what == "ply":
                env.ply_repeat(n, R, feat_out=feat, legal_out=legal)
            else:
                for _ in range(n):
                    env.features(feat, None, legal)
        run(3)
        torch.cuda.synchronize()
        env.timer_begin()
        run(N)
        env.timer_end()
        out.append(env.timer_ms() * 1e3 / N)
    print("%.2f %.2f" % tuple(out), flush=True)
    sys.exit(0)
B = int(sys.argv[1])
for pad in [0] + [int(x) for x in sys.argv[2:]]:
    env = dict(os.environ, GOENV_LDS_PAD=str(pad))
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B)], capture_output=True, text=True, timeout=300, env=env)
    print("%6d boards, LDS pad %6d B per workgroup: k_ply %s us, k_features %s us" % ((B, pad) + tuple((r.stdout.split() or ["?", r.stderr[-300:]])[-2:])), flush=True)
