#!/usr/bin/env python3
"""Golden vectors for the tree search: the UNMODIFIED reference `go_game` module
(oracle/_ref/go_game*.so, `make -C oracle refmod`) is run in selfplay mode with use_mcts and ONE
search thread per game (the only setting in which the reference is deterministic), the policy /
value "network" being tests/common.py:mcts_policy.  Recorded per game: the stream of leaf
evaluations it asks for -- (seq, game_counter, last_terminal, move_idx, winner, aug_code,
game_record_idx, FNV of the planes) -- which fixes the order of expansion, every position the
search visits and every move it plays.

    python tests/golden/make_golden_mcts.py        -> tests/golden/mcts_stream.json
"""
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import GOLDEN, dump_json, mcts_policy  # noqa: E402
from oracle.pyoracle import FNV_SEED, fnv_planes  # noqa: E402

RUNS = [
    dict(name="plain", seed=11, num_games=1, rollouts=12, use_prior=False, pseudo_games=0, pick="most_visited", evals=130),
    dict(name="prior", seed=5, num_games=2, rollouts=20, use_prior=True, pseudo_games=0, pick="most_visited", evals=130),
    dict(name="pseudo_strongest", seed=9, num_games=1, rollouts=8, use_prior=True, pseudo_games=3,
         pick="strongest_prior", evals=90),
]
NP_TYPES = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}
IN_KEYS = ["s", "id", "seq", "game_counter", "last_terminal", "move_idx", "winner", "aug_code", "game_record_idx"]
_KEEP_ALIVE = []


def run_reference(go, cfg):
    co = go.ContextOptions()
    co.num_games = cfg["num_games"]
    co.T = 1
    co.num_collectors = 1
    co.max_num_threads = 1
    ts = co.mcts_options
    ts.num_threads = 1
    ts.num_rollout_per_thread = cfg["rollouts"]
    ts.use_prior = cfg["use_prior"]
    ts.pseudo_games = cfg["pseudo_games"]
    ts.pick_method = cfg["pick"]
    ts.persistent_tree = False
    ts.verbose = False
    opt = go.GameOptions()
    opt.seed = cfg["seed"]
    opt.mode = "selfplay"
    opt.use_mcts = True
    opt.verbose = False
    GC = go.GameContext(co, opt)
    gstat = GC.CreateGroupStat()
    gstat.hist_len = 1
    gstat.name = "actor"
    B = cfg["num_games"]
    gid = GC.AddCollectors(B, 0, 0, gstat)
    bufs = {}
    for which, keys in (("input", IN_KEYS), ("reply", ["pi", "V"])):
        for key in keys:
            info = GC.GetTensorSpec(gid, key, 1)
            assert info.key == key, key
            v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
            info.p = v.ctypes.data
            info.byte_size = v.size * v.dtype.itemsize
            GC.AddTensor(gid, which, info)
            bufs[key] = v
    GC.Start()
    streams = {g: [] for g in range(B)}
    while min(len(v) for v in streams.values()) < cfg["evals"]:
        infos = GC.Wait(0)
        for r in range(len(infos.s)):
            g = int(bufs["id"][0, r])
            s = bufs["s"][0, r]
            streams[g].append([int(bufs[k][0, r]) for k in IN_KEYS[2:]] + ["%016x" % fnv_planes(FNV_SEED, s.reshape(-1))])
            pi, v = mcts_policy(s)
            bufs["pi"][0, r] = pi
            bufs["V"][0, r] = v
        GC.Steps(infos)
    _KEEP_ALIVE.append((GC, bufs))   # the reference's teardown blocks: the child leaves with os._exit
    return {str(g): v[:cfg["evals"]] for g, v in streams.items()}


def main():
    so = glob.glob(os.path.join(ROOT, "oracle", "_ref", "go_game*.so"))
    if not so:
        raise SystemExit("oracle/_ref/go_game*.so missing: run `make -C oracle refmod` where /root/reference is mounted")
    if len(sys.argv) > 2 and sys.argv[1] == "--run":
        sys.path.insert(0, os.path.dirname(so[0]))
        import go_game as go
        st = run_reference(go, RUNS[int(sys.argv[2])])
        dump_json(sys.argv[3], st)
        sys.stdout.flush()
        os._exit(0)
    import subprocess
    import tempfile
    out = dict(fields=IN_KEYS[2:] + ["s_fnv"], runs=[])
    for i, cfg in enumerate(RUNS):
        with tempfile.TemporaryDirectory() as td:
            tmp = os.path.join(td, "run.json")
            for attempt in range(4):
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--run", str(i), tmp],
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
                if r.returncode == 0 and os.path.exists(tmp):
                    break
            else:
                raise SystemExit("reference run %d failed" % i)
            with open(tmp) as f:
                st = json.load(f)
        out["runs"].append(dict(cfg=cfg, streams=st))
        print(cfg["name"], {g: (len(v), v[-1][3]) for g, v in st.items()}, flush=True)
    dump_json(os.path.join(GOLDEN, "mcts_stream.json"), out)


if __name__ == "__main__":
    main()
