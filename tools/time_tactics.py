#!/usr/bin/env python3
"""Time goenv_tactics (heuristic move filters) on midgame positions, and beside it the reference's own
functions (IsSelfAtari, IsMoveGivingSimpleKo, FindAllCandidateMoves, find_*_libert*, CheckLadder over all
361 points: oracle/_ref harness, one host thread) on a sample of the same positions.
Usage: tools/time_tactics.py [boards] [plies]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time  # noqa: E402

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
    cpu_reference(plies)


def cpu_reference(plies, sample=48):
    """The same positions (boards 0..sample-1 of the same seed) through the reference build."""
    from oracle.pyoracle import OracleBoard, RefBoard
    cls, kind = (RefBoard, "reference build") if RefBoard.available() else (OracleBoard, "C port")
    boards = []
    for b in range(sample):
        o = cls()
        for _ in range(plies):
            if o.is_game_end():
                break
            o.forward(o.policy_move(3, b, 0))
        boards.append(o)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        for o in boards:
            o.tactics(2)
    dt = (time.perf_counter() - t0) / (reps * sample)
    print("CPU %s, all five filters over 361 points, one thread: %.1f us/board (%d boards sampled)" % (kind, dt * 1e6, sample))


if __name__ == "__main__":
    main()
