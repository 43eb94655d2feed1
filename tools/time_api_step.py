#!/usr/bin/env python3
"""Where a fused selfplay step (goenv_ply_policy_sel) spends its time against the launch-per-ply
kernel with the on-device policy: same boards, same outputs, Python-issued launches timed with HIP
events.  Usage: tools/time_api_step.py [boards] [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from elf_amd import goenv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400


def timed(env, fn, n=N, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    env.timer_begin()
    for _ in range(n):
        fn()
    env.timer_end()
    return env.timer_ms() / n * 1e3


def main():
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    pi = torch.ones(B, 361, device=env.device)
    R = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    E = goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END
    env.ply_repeat(100, R)
    out = {}
    out["ply random+auto_restart"] = timed(env, lambda: env.ply(R, feat_out=feat, legal_out=legal))
    out["ply random+restart_after_end"] = timed(env, lambda: env.ply(goenv.F_RANDOM_POLICY | goenv.F_RESTART_AFTER_END,
                                                                     feat_out=feat, legal_out=legal))
    out["policy_sel uniform pi (midgame boards)"] = timed(env, lambda: env.ply_policy_sel(pi, flags=E, feat_out=feat,
                                                                                         legal_out=legal), n=60, warm=0)
    out["policy_sel uniform pi (after 600 more plies)"] = timed(
        env, lambda: env.ply_policy_sel(pi, flags=E, feat_out=feat, legal_out=legal), warm=600)
    out["ply random on those boards"] = timed(env, lambda: env.ply(R, feat_out=feat, legal_out=legal), n=60, warm=0)
    out["policy_sel, no planes"] = timed(env, lambda: env.ply_policy_sel(pi, flags=E, legal_out=legal))
    out["ply random, no planes"] = timed(env, lambda: env.ply(R, legal_out=legal))
    ext = torch.full((B,), -1, dtype=torch.int64, device=env.device)
    out["ply external pass, planes"] = timed(env, lambda: env.ply(goenv.F_AUTO_RESTART, actions=ext, feat_out=feat, legal_out=legal))
    for nw in (1, 2, 3, 6):
        env.set_waves_per_board(nw)
        out["policy_sel planes, %d waves/board" % nw] = timed(env, lambda: env.ply_policy_sel(pi, flags=E, feat_out=feat, legal_out=legal), n=100)
        out["ply random planes, %d waves/board" % nw] = timed(env, lambda: env.ply(R, feat_out=feat, legal_out=legal), n=100)
    env.set_waves_per_board(0)
    acts = torch.zeros(B, dtype=torch.int64, device=env.device)
    out["sample_policy alone"] = timed(env, lambda: env.sample_policy(pi, acts))
    st = env.get_state()
    stones = (st["color"] != 0).sum(1).float()
    print("boards %d: mean stones %.1f, mean ply %.1f" % (B, stones.mean().item(), st["scalars"][:, 0].float().mean().item()))
    for k, v in out.items():
        print("%-50s %8.2f us" % (k, v))


if __name__ == "__main__":
    main()
