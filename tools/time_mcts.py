#!/usr/bin/env python3
"""Tree search behind the GameContext (use_mcts): leaf evaluations per second with a
policy/value answer that costs nothing (a fixed `pi`, `V` = 0.5, written once into the reply tensors),
so that what is timed is the search itself: descending the tree, making the leaf positions, extracting
their planes into the batch, taking the five best legal points of `pi`, expanding, backing up, playing.

  ours        elf_amd.go_game: host trees advancing in lockstep, positions and planes on the GPU, one
              batch per Wait with the parked leaves of ALL games (up to games x threads rows).
  reference   the UNMODIFIED reference go_game module (oracle/_ref): `threads` OS threads per game,
              each blocking on the collector for one leaf; numpy buffers.  Child process (os._exit).

Usage: tools/time_mcts.py [--games 256] [--threads 16] [--rollouts 100] [--evals 200000]
                          [--ref-games 16] [--ref-evals 20000] [--skip-reference]"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NP_TYPES = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}


def make(go, games, threads, rollouts, batch, make_buffer, timeout_usec=0):
    co = go.ContextOptions()
    co.num_games = games
    co.T = 1
    co.num_collectors = 1
    co.max_num_threads = threads
    ts = co.mcts_options
    ts.num_threads, ts.num_rollout_per_thread = threads, rollouts
    ts.use_prior, ts.pseudo_games, ts.pick_method = True, 0, "most_visited"
    ts.persistent_tree, ts.verbose = False, False
    opt = go.GameOptions()
    opt.seed, opt.mode, opt.use_mcts, opt.verbose = 5, "selfplay", True, False
    gc = go.GameContext(co, opt)
    gs = gc.CreateGroupStat()
    gs.hist_len, gs.name = 1, "actor"
    gid = gc.AddCollectors(batch, 0, timeout_usec, gs)
    bufs = {}
    for which, keys in (("input", ["s"]), ("reply", ["pi", "V"])):
        for key in keys:
            info = gc.GetTensorSpec(gid, key, 1)
            v, info.p, info.byte_size = make_buffer(info)
            gc.AddTensor(gid, which, info)
            bufs[key] = v
    return gc, bufs


def fixed_pi():
    rs = np.random.RandomState(1)
    pi = (rs.permutation(361) + 1).astype(np.float32)
    return pi / pi.sum()


def loop(gc, evals, warm):
    n, n0, t0, batches = 0, None, None, 0
    while n0 is None or n - n0 < evals:
        if n0 is None and n >= warm:
            n0, t0 = n, time.perf_counter()
        infos = gc.Wait(0)
        n += len(infos.s)
        if n0 is not None:
            batches += 1
        gc.Steps(infos)
    return n - n0, time.perf_counter() - t0, batches


def run_ours(args):
    import torch
    from elf_amd import go_game
    tt = {"float": torch.float32}

    def dev_buffer(info):
        v = torch.zeros(*info.sz, dtype=tt[info.type], device="cuda:0")
        return v, v.data_ptr(), v.numel() * v.element_size()

    gc, bufs = make(go_game, args.games, args.threads, args.rollouts, args.games * args.threads, dev_buffer)
    bufs["pi"][:] = torch.from_numpy(fixed_pi()).cuda()
    bufs["V"].fill_(0.5)
    gc.Start()
    n, dt, batches = loop(gc, args.evals, args.evals // 10)
    torch.cuda.synchronize()
    moves = sum(gc.SearchInfo(g)["moves_played"] for g in range(args.games))
    ht = gc.HostTimes()
    gc.Stop()
    return {"games": args.games, "threads": args.threads, "rollouts_per_thread": args.rollouts, "evaluations": n,
            "seconds": dt, "evaluations_per_s": n / dt, "rows_per_batch": n / max(1, batches), "moves_played": moves,
            "us_per_batch": dt / max(1, batches) * 1e6,
            "host_us_per_batch": (ht["wait_us"] + ht["steps_us"]) / max(1, ht["waits"]),
            "host_wait_us_per_batch": ht["wait_us"] / max(1, ht["waits"]), "host_steps_us_per_batch": ht["steps_us"] / max(1, ht["steps"]),
            "host_threads": int(os.environ.get("GOENV_HOST_THREADS", "0")) or min(16, len(os.sched_getaffinity(0)))}


def run_reference_child(args):
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so"))
    sys.path.insert(0, os.path.dirname(so[0]))
    import go_game as go

    def host_buffer(info):
        v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
        return v, v.ctypes.data, v.size * v.dtype.itemsize

    batch = max(1, args.ref_games * args.threads // 4)
    gc, bufs = make(go, args.ref_games, args.threads, args.rollouts, batch, host_buffer, timeout_usec=10)
    bufs["pi"][:] = fixed_pi()
    bufs["V"][:] = 0.5
    gc.Start()
    try:
        n, dt, batches = loop(gc, args.ref_evals, args.ref_evals // 10)
    except BaseException as e:      # the reference's teardown blocks: never fall out of this function
        print(json.dumps({"error": repr(e)}), flush=True)
        os._exit(1)
    print(json.dumps({"games": args.ref_games, "threads": args.threads, "rollouts_per_thread": args.rollouts,
                      "os_threads": args.ref_games * args.threads, "cpus": len(os.sched_getaffinity(0)),
                      "evaluations": n, "seconds": dt, "evaluations_per_s": n / dt,
                      "rows_per_batch": n / max(1, batches)}), flush=True)
    os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--rollouts", type=int, default=100)
    ap.add_argument("--evals", type=int, default=200000)
    ap.add_argument("--ref-games", type=int, default=16)
    ap.add_argument("--ref-evals", type=int, default=20000)
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--reference-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.reference_child:
        run_reference_child(args)
    out = {}
    if not args.skip_reference and glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reference-child", "--ref-games",
                            str(args.ref_games), "--threads", str(args.threads), "--rollouts", str(args.rollouts),
                            "--ref-evals", str(args.ref_evals)], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["reference_cpu"] = json.loads(lines[-1]) if lines else {"error": (r.stderr or r.stdout)[-400:]}
    out["ours"] = run_ours(args)
    if "evaluations_per_s" in out.get("reference_cpu", {}):
        out["ratio"] = out["ours"]["evaluations_per_s"] / out["reference_cpu"]["evaluations_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
