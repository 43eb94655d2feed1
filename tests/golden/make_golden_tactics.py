#!/usr/bin/env python3
"""Golden vectors for the heuristic move filters, recorded from the UNMODIFIED
reference (oracle/_ref/libgoref.so -> go/board.cc IsSelfAtari, IsMoveGivingSimpleKo,
FindAllCandidateMoves, find_only_liberty, find_two_liberties, CheckLadder).  Run in the build
container (needs /root/reference only through the prebuilt oracle/_ref):

    python tests/golden/make_golden_tactics.py        -> tests/golden/tactics.json

Positions: three corpus records, every 8th position plus every second one that has a ladder
move, self-atari thresholds 1 / 3 / 100.  Each
position stores the moves that lead to it and the outputs in sparse form."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np  # noqa: E402

from common import dump_json  # noqa: E402
from oracle.pyoracle import RefBoard  # noqa: E402


def sparse(a):
    a = np.asarray(a)
    idx = np.flatnonzero(a.reshape(361, -1).any(1) if a.ndim > 1 else a)
    return [[int(i), a[i].tolist()] for i in idx]


def main():
    out = []
    for name in ("plain0.sgf", "handi5.sgf", "sample1.sgf"):
        path = os.path.join(HERE, "corpus", name)
        rec = RefBoard.sgf_load(path)
        r = RefBoard()
        r.apply_handicap(rec["handi"])
        moves = []
        for i, m in enumerate(rec["moves"]):
            ok, _ = r.forward(int(m))
            if not ok:
                break
            moves.append(int(m))
            thres = [1, 3, 100][(i // 8) % 3]
            t = r.tactics(thres)
            if i % 8 == 7 or (t["ladder"].max() > 0 and i % 2 == 0):
                libs = t["libs"] + 1   # -1 -> 0 so that sparse() drops "none"
                out.append(dict(record=name, handi=int(rec["handi"]), n_moves=len(moves), thres=thres,
                                self_atari=sparse(t["self_atari"]), gives_ko=sparse(t["gives_ko"]),
                                not_cand=sparse(1 - t["cand"]), libs_plus1=sparse(libs), ladder=sparse(t["ladder"])))
    records = {}
    for name in ("plain0.sgf", "handi5.sgf", "sample1.sgf"):
        rec = RefBoard.sgf_load(os.path.join(HERE, "corpus", name))
        records[name] = dict(handi=int(rec["handi"]), moves=[int(m) for m in rec["moves"]])
    dump_json(os.path.join(HERE, "tactics.json"), dict(records=records, positions=out))
    print(len(out), "positions,", sum(len(p["ladder"]) for p in out), "ladder moves")


if __name__ == "__main__":
    main()
