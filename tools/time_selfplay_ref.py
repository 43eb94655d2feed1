#!/usr/bin/env python3
"""ELF's OWN selfplay path on the host CPUs (north star: "next to ELF's own ctpl-threaded CPU path, timed on
the same box's host cores"): the UNMODIFIED reference go_game module (oracle/_ref, built from /root/reference by
`make -C oracle refmod`) in selfplay mode -- one OS thread per game, DirectPredictAI::Act = BoardFeature::Extract
+ the collector hand-shake + GoState::forward per move (go/go_ai.h:79-93, elf/comm_template.h:215-261), batches
collected into numpy buffers.  The reply is a move `a` per sample: a random EMPTY point read off plane 9 of the
batch (nearly always legal; the reference restarts a game on an illegal move, go/game.cc:81-90), computed with a
few vectorised numpy operations so that the policy costs next to nothing.  Runs in a child process that leaves
with os._exit (the reference's teardown blocks), stdout of the module itself (it prints a board per illegal
move) sent to /dev/null.

Usage: tools/time_selfplay_ref.py [--games 1024] [--batch 1024] [--batches 60]"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NP_TYPES = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}


def child(args):
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so"))
    sys.path.insert(0, os.path.dirname(so[0]))
    out_fd = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)                                  # the module's own prints
    import go_game as go
    try:
        co = go.ContextOptions()
        co.num_games, co.T, co.num_collectors, co.max_num_threads = args.games, 1, 1, 0
        opt = go.GameOptions()
        opt.seed, opt.mode, opt.use_mcts, opt.verbose = 7, "selfplay", False, False
        gc = go.GameContext(co, opt)
        gs = gc.CreateGroupStat()
        gs.hist_len, gs.name = 1, "actor"
        gid = gc.AddCollectors(args.batch, 0, 10, gs)
        bufs = {}
        for which, keys in (("input", ("s",)), ("reply", ("pi", "V", "a"))):
            for key in keys:
                info = gc.GetTensorSpec(gid, key, 1)
                v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
                info.p, info.byte_size = v.ctypes.data, v.size * v.dtype.itemsize
                gc.AddTensor(gid, which, info)
                bufs[key] = v
        prio = np.random.RandomState(1).rand(args.batch, 361).astype(np.float32) + 0.5
        gc.Start()
        n, t0, warm = 0, None, max(4, args.batches // 10)
        for i in range(warm + args.batches):
            if i == warm:
                t0, n = time.perf_counter(), 0
            infos = gc.Wait(0)
            k = len(infos.s)
            empty = bufs["s"][0, :k, 9].reshape(k, 361)
            bufs["a"][0, :k] = np.argmax(empty * prio[:k], axis=1)
            n += k
            gc.Steps(infos)
        dt = time.perf_counter() - t0
        res = {"games": args.games, "os_threads": args.games, "batch": args.batch, "samples": n, "seconds": dt,
               "moves_per_s": n / dt, "cpus": len(os.sched_getaffinity(0))}
    except BaseException as e:
        res = {"error": repr(e)}
    os.write(out_fd, (json.dumps(res) + "\n").encode())
    os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--batches", type=int, default=60)
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child:
        child(args)
    if not glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so")):
        print(json.dumps({"error": "oracle/_ref/go_game*.so not built"}))
        return
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--games", str(args.games), "--batch",
                        str(args.batch), "--batches", str(args.batches)], capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    print(lines[-1] if lines else json.dumps({"error": (r.stderr or r.stdout)[-400:]}))


if __name__ == "__main__":
    main()
