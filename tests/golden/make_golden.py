"""Generates the committed golden vectors from the UNMODIFIED reference
(oracle/_ref/libgoref.so, built by `make -C oracle` from /root/reference).

Run here (the container that mounts /root/reference):  python tests/golden/make_golden.py
The outputs are data only -- move lists, legality flags, digests and a few full
arrays -- never reference source.

Files written (tests/golden/):
  sample1.json    replay of go/sample1.sgf (the reference's only Go fixture): parsed moves,
                  header fields, per-ply position digests (aug 0), final known answers
  playouts.json   cfg-2 policy playouts on the reference board: moves + per-ply digests
  edge.json       hand-built edge cases (tests/common.py EDGE_CASES): per-move legality,
                  digests, scores
  handicap.json   handicap 0..9: digests + stone lists
  d4.json         Coord2Action/Action2Coord tables for the 8 D4 codes + plane digests
  planes_full.npz full planes / canonical arrays of a handful of positions
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from common import EDGE_CASES, GOLDEN, board_digest, dump_json  # noqa: E402
from oracle.pyoracle import FNV_SEED, RefBoard, fnv_planes  # noqa: E402

SEED = 20251110


def gen_sample1():
    path = os.path.join(GOLDEN, "sample1.sgf")
    sg = RefBoard.sgf_load(path)
    b = RefBoard()
    digests, legal_flags = [], []
    h = FNV_SEED
    captures = kos = 0
    for i in range(sg["num_moves"]):
        digests.append(board_digest(b))
        h = fnv_planes(h, b.extract(0))
        before = b.canonical()["scalars"].copy()
        ok, _ = b.forward(int(sg["moves"][i]))
        legal_flags.append(int(ok))
        after = b.canonical()["scalars"]
        if after[2] + after[3] > before[2] + before[3]:
            captures += 1
        if after[8] == 0:
            kos += 1
    digests.append(board_digest(b))
    c = b.canonical()
    out = dict(
        moves=[int(m) for m in sg["moves"]], players=[int(p) for p in sg["players"]],
        num_moves=sg["num_moves"], winner=sg["winner"], handi=sg["handi"], board_size=sg["board_size"],
        legal=legal_flags, digests=digests, planes_fnv="%016x" % h,
        final=dict(scalars=[int(v) for v in c["scalars"]], stones=int((c["color"] > 0).sum()),
                   num_groups=b.num_groups(), tt_score=b.tt_score(), fast_score_cn=b.fast_score(0),
                   fast_score_jp=b.fast_score(1), capturing_moves=captures, ko_moves=kos))
    dump_json(os.path.join(GOLDEN, "sample1.json"), out)
    return b


def gen_playouts(n_boards=12, max_ply=722):
    games = []
    full = {}
    for bid in range(n_boards):
        b = RefBoard()
        moves, digests = [], []
        while True:
            digests.append(board_digest(b, aug=bid % 8))
            a = b.policy_move(SEED, bid, 0)
            ok, end = b.forward(a)
            assert ok
            moves.append(a)
            if end or b.canonical()["scalars"][0] > max_ply:
                break
        digests.append(board_digest(b, aug=bid % 8))
        games.append(dict(board_id=bid, aug=bid % 8, moves=moves, digests=digests, tt_score=b.tt_score(),
                          fast_score_cn=b.fast_score(0), final_scalars=[int(v) for v in b.canonical()["scalars"]]))
        if bid < 3:
            c = b.canonical()
            for k in ("color", "liberties", "stones", "last_placed", "scalars"):
                full["playout%d_%s" % (bid, k)] = c[k]
            full["playout%d_planes" % bid] = b.extract(bid % 8)
            full["playout%d_legal" % bid] = b.legal()
    dump_json(os.path.join(GOLDEN, "playouts.json"), dict(seed=SEED, max_ply=max_ply, games=games))
    return full


def gen_edge():
    out = {}
    for name, seq in EDGE_CASES.items():
        b = RefBoard()
        legal, ends, digests, tt, fast = [], [], [], [], []
        for a in seq:
            ok, end = b.forward(a)
            legal.append(int(ok))
            ends.append(int(end) if ok else 0)
            digests.append(board_digest(b))
            tt.append(b.tt_score())
            fast.append(b.fast_score(0))
        out[name] = dict(moves=seq, legal=legal, game_end=ends, digests=digests, tt_score=tt, fast_score_cn=fast,
                         final_scalars=[int(v) for v in b.canonical()["scalars"]])
    dump_json(os.path.join(GOLDEN, "edge.json"), out)


def gen_handicap():
    out = {}
    for h in range(0, 11):
        b = RefBoard()
        b.apply_handicap(h)
        c = b.canonical()
        # a few moves on top so the "white moves first" consequence is pinned too
        after = []
        for a in (60, 300, 180):
            ok, _ = b.forward(a)
            after.append(dict(action=a, legal=int(ok), digest=board_digest(b)))
        out[str(h)] = dict(stones=[int(p) for p in np.flatnonzero(c["color"] == 1)],
                           scalars=[int(v) for v in c["scalars"]], after=after)
    dump_json(os.path.join(GOLDEN, "handicap.json"), out)


def gen_d4(b):
    out = dict(coord2action=[[RefBoard.aug_action(code, a) for a in range(361)] for code in range(8)],
               action2coord=[[RefBoard.aug_action_inv(code, a) for a in range(361)] for code in range(8)],
               action2coord_pass=[RefBoard.aug_action_inv(code, -1) for code in range(8)],
               sample1_final_digests=[board_digest(b, aug=code) for code in range(8)])
    dump_json(os.path.join(GOLDEN, "d4.json"), out)


def main():
    if not RefBoard.available():
        raise SystemExit("oracle/_ref/libgoref.so missing: run `make -C oracle` where /root/reference is mounted")
    b = gen_sample1()
    full = gen_playouts()
    gen_edge()
    gen_handicap()
    gen_d4(b)
    c = b.canonical()
    for k in ("color", "liberties", "stones", "last_placed", "scalars"):
        full["sample1_final_%s" % k] = c[k]
    for code in range(8):
        full["sample1_final_planes_aug%d" % code] = b.extract(code)
    full["sample1_final_legal"] = b.legal()
    np.savez_compressed(os.path.join(GOLDEN, "planes_full.npz"), **full)
    print("golden vectors written to", GOLDEN)


if __name__ == "__main__":
    main()
