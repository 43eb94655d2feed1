#!/usr/bin/env python3
"""cfg 3 (BASELINE.json configs[2]): offline SGF replay -> batches of 256 samples (25 planes +
offline_a + labels), samples per second.

  ours        elf_amd.go_game.GameContext in offline mode: records, iterators and RNG streams on the
              host as OfflineLoader keeps them, every rules operation and the planes on the GPU,
              batch tensors in HBM (or host buffers through a staging copy with --host).
  reference   the UNMODIFIED reference go_game module (oracle/_ref, built from /root/reference by
              `make -C oracle refmod`): one OS thread per game, collector memcpy into numpy buffers.
              Run as a child process that leaves with os._exit (its teardown blocks).

Same corpus (tests/golden/corpus/train.lst), same GameOptions.  Usage:
    tools/time_offline.py [--games 256] [--batches 200] [--our-batches 4000] [--host] [--skip-reference]"""
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
CORPUS = os.path.join(ROOT, "tests", "golden", "corpus")
NP_TYPES = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}
KEYS = ["s", "offline_a", "id", "seq", "game_counter", "last_terminal", "move_idx", "winner", "aug_code",
        "game_record_idx"]


def options(go, games):
    co = go.ContextOptions()
    co.num_games = games
    co.T = 1
    co.num_collectors = 1
    opt = go.GameOptions()
    opt.seed, opt.mode = 7, "offline"
    opt.list_filename = os.path.join(CORPUS, "train.lst")
    opt.data_aug, opt.start_ratio_pre_moves, opt.ratio_pre_moves = -1, 0.5, 0.0
    opt.move_cutoff, opt.num_games_per_thread, opt.num_future_actions = -1, 300, 3
    opt.verbose = False
    return co, opt


def drive(go, games, batches, make_buffer, sync, warm=20):
    co, opt = options(go, games)
    gc = go.GameContext(co, opt)
    gs = gc.CreateGroupStat()
    gs.hist_len = 1
    gid = gc.AddCollectors(games, 0, 0, gs)
    bufs = {}
    for key in KEYS:
        info = gc.GetTensorSpec(gid, key, 1)
        v, info.p, info.byte_size = make_buffer(info)
        gc.AddTensor(gid, "input", info)
        bufs[key] = v
    gc.Start()
    n = 0
    for i in range(warm + batches):
        if i == warm:
            sync()
            if hasattr(gc, "HostTimes"):
                gc.HostTimes(True)
            t0, n = time.perf_counter(), 0
        infos = gc.Wait(0)
        n += len(infos.s)
        gc.Steps(infos)
    sync()
    dt = time.perf_counter() - t0
    return gc, bufs, n, dt


def run_ours(args):
    import torch
    from elf_amd import go_game
    tt = {"int32_t": torch.int32, "int64_t": torch.int64, "float": torch.float32, "unsigned char": torch.uint8,
          "char": torch.int8}

    def dev_buffer(info):
        v = torch.zeros(*info.sz, dtype=tt[info.type], device="cuda:0")
        return v, v.data_ptr(), v.numel() * v.element_size()

    def host_buffer(info):
        v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
        return v, v.ctypes.data, v.nbytes

    batches = args.our_batches if not args.host else min(args.our_batches, 400)
    gc, bufs, n, dt = drive(go_game, args.games, batches, host_buffer if args.host else dev_buffer,
                            torch.cuda.synchronize)
    ht = gc.HostTimes()
    gc.Stop()
    return {"samples": n, "seconds": dt, "samples_per_s": n / dt, "tensors": "host" if args.host else "device",
            "batches": batches, "us_per_batch": dt / batches * 1e6, "host_wait_us_per_batch": ht["wait_us"] / max(1, ht["waits"]),
            "host_steps_us_per_batch": ht["steps_us"] / max(1, ht["steps"])}


def run_reference_child(args):
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so"))
    sys.path.insert(0, os.path.dirname(so[0]))
    import go_game as go

    def host_buffer(info):
        v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
        return v, v.ctypes.data, v.size * v.dtype.itemsize

    try:
        keep = drive(go, args.games, args.batches, host_buffer, lambda: None)
    except BaseException as e:      # the reference's teardown blocks: never fall out of this function
        print(json.dumps({"error": repr(e)}), flush=True)
        os._exit(1)
    print(json.dumps({"samples": keep[2], "seconds": keep[3], "samples_per_s": keep[2] / keep[3],
                      "threads": args.games, "cpus": len(os.sched_getaffinity(0))}), flush=True)
    os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=256)
    ap.add_argument("--batches", type=int, default=200, help="timed batches of the reference module")
    ap.add_argument("--our-batches", type=int, default=4000, help="timed batches of this build (16 us each: 200 would be 3 ms)")
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--skip-reference", action="store_true")
    ap.add_argument("--reference-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.reference_child:
        run_reference_child(args)
    out = {"config": "cfg3: offline SGF replay, %d games, batch %d, corpus train.lst (13 records), k=3, random D4"
                     % (args.games, args.games)}
    if not args.skip_reference and glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so")):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reference-child", "--games", str(args.games),
                            "--batches", str(args.batches)], capture_output=True, text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["reference_cpu"] = json.loads(lines[-1]) if lines else {"error": r.stderr[-400:]}
    out["ours"] = run_ours(args)
    if "samples_per_s" in out.get("reference_cpu", {}):
        out["ratio"] = out["ours"]["samples_per_s"] / out["reference_cpu"]["samples_per_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
