#!/usr/bin/env python3
"""Time goenv_tactics (heuristic move filters) on midgame positions.  Usage: tools/time_tactics.py [boards] [plies]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from elf_amd import goenv  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    plies = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    env = goenv.GoEnv(B, seed=3, max_ply=722)
    env.ply_repeat(plies, goenv.F_RANDOM_POLICY)
    dev = env.device
    outs = dict(self_atari=torch.zeros((B, 361), dtype=torch.int16, device=dev),
                gives_ko=torch.zeros((B, 361), dtype=torch.uint8, device=dev),
                cand=torch.zeros((B, 361), dtype=torch.uint8, device=dev),
                libs=torch.zeros((B, 361, 2), dtype=torch.int16, device=dev),
                ladder=torch.zeros((B, 361), dtype=torch.int16, device=dev))
    for name, sel in (("all five", outs), ("without ladder", {k: v for k, v in outs.items() if k != "ladder"}),
                      ("cand only", {"cand": outs["cand"]})):
        env.tactics(2, **sel)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            env.tactics(2, **sel)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%6d boards after %3d plies, %-15s %9.1f us/launch  %7.3f us/board" % (B, plies, name, us, us / B))
    print("ladder moves per board: %.2f, self-atari moves per board: %.2f" % (
        float((outs["ladder"] > 0).sum()) / B, float((outs["self_atari"] > 0).sum()) / B))


if __name__ == "__main__":
    main()
