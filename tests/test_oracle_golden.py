"""CPU: the plain-C oracle (oracle/go_oracle.c) against the golden vectors generated from the
unmodified reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from common import EDGE_CASES, GOLDEN, assert_canon_equal, assert_planes_equal, board_digest, load_golden
from oracle.pyoracle import FNV_SEED, OracleBoard, fnv_planes


def test_sample1_replay_known_answers():
    g = load_golden("sample1.json")
    assert g["num_moves"] == 258 and len(g["moves"]) == 260 and g["winner"] == 1 and g["handi"] == 0
    b = OracleBoard()
    h = FNV_SEED
    for i in range(g["num_moves"]):
        assert board_digest(b) == g["digests"][i], "ply %d" % i
        h = fnv_planes(h, b.extract(0))
        ok, _ = b.forward(g["moves"][i])
        assert ok == bool(g["legal"][i])
    assert board_digest(b) == g["digests"][-1]
    assert "%016x" % h == g["planes_fnv"]
    c = b.canonical()
    f = g["final"]
    assert [int(v) for v in c["scalars"]] == f["scalars"]
    # SURVEY 8c known answers of the reference replay
    assert c["scalars"][0] == 259 and c["scalars"][2] == 33 and c["scalars"][3] == 4
    assert int((c["color"] > 0).sum()) == 221 == f["stones"]
    assert b.tt_score() == 88.0 == f["tt_score"]
    assert b.fast_score(0) == 25.0 == f["fast_score_cn"]
    assert b.fast_score(1) == f["fast_score_jp"]


def test_full_arrays_of_sample1_final_position():
    g = load_golden("sample1.json")
    z = np.load(os.path.join(GOLDEN, "planes_full.npz"))
    b = OracleBoard()
    for a in g["moves"][:g["num_moves"]]:
        b.forward(a)
    c = b.canonical()
    ref = {k: z["sample1_final_" + k] for k in ("color", "liberties", "stones", "last_placed", "scalars")}
    assert_canon_equal(c, ref, "sample1 final")
    assert np.array_equal(b.legal(), z["sample1_final_legal"])
    for code in range(8):
        assert_planes_equal(b.extract(code), z["sample1_final_planes_aug%d" % code], "aug %d" % code)


def test_policy_playouts_match_reference():
    g = load_golden("playouts.json")
    for game in g["games"]:
        b = OracleBoard()
        for i, a in enumerate(game["moves"]):
            assert board_digest(b, game["aug"]) == game["digests"][i], (game["board_id"], i)
            assert b.policy_move(g["seed"], game["board_id"], 0) == a
            ok, _ = b.forward(a)
            assert ok
        assert board_digest(b, game["aug"]) == game["digests"][-1]
        assert b.tt_score() == game["tt_score"] and b.fast_score(0) == game["fast_score_cn"]
        # the C playout driver produces the same sequence
        b2 = OracleBoard()
        assert b2.playout(g["seed"], game["board_id"], 0, g["max_ply"]).tolist() == game["moves"]


@pytest.mark.parametrize("name", sorted(EDGE_CASES))
def test_edge_cases(name):
    g = load_golden("edge.json")[name]
    assert g["moves"] == EDGE_CASES[name]
    b = OracleBoard()
    for i, a in enumerate(g["moves"]):
        ok, end = b.forward(a)
        assert int(ok) == g["legal"][i], (name, i)
        if ok:
            assert int(end) == g["game_end"][i]
        assert board_digest(b) == g["digests"][i], (name, i)
        assert b.tt_score() == g["tt_score"][i] and b.fast_score(0) == g["fast_score_cn"][i]


def test_handicap_table():
    g = load_golden("handicap.json")
    for h in range(0, 11):
        b = OracleBoard()
        b.apply_handicap(h)
        c = b.canonical()
        e = g[str(h)]
        assert [int(p) for p in np.flatnonzero(c["color"] == 1)] == e["stones"]
        assert [int(v) for v in c["scalars"]] == e["scalars"]
        for step in e["after"]:
            ok, _ = b.forward(step["action"])
            assert int(ok) == step["legal"] and board_digest(b) == step["digest"]


def test_d4_tables():
    g = load_golden("d4.json")
    for code in range(8):
        assert [OracleBoard.aug_action(code, a) for a in range(361)] == g["coord2action"][code]
        assert [OracleBoard.aug_action_inv(code, a) for a in range(361)] == g["action2coord"][code]
        assert OracleBoard.aug_action_inv(code, -1) == g["action2coord_pass"][code] == -1
    # SURVEY appendix worked example: "pd" = (15,3)
    assert [g["coord2action"][c][15 * 19 + 3] for c in range(8)] == [288, 60, 72, 300, 72, 60, 288, 300]


def test_empty_board_and_set_state_roundtrip():
    b = OracleBoard()
    assert b.tt_score() == 0.0 and b.legal().sum() == 361 and b.candidates().sum() == 361
    g = load_golden("playouts.json")["games"][0]
    for a in g["moves"][:200]:
        b.forward(a)
    c = b.canonical()
    b2 = OracleBoard()
    b2.set_state(c["color"], c["last_placed"], c["scalars"])
    assert board_digest(b2) == board_digest(b)


def test_move_filters_golden():
    """Heuristic move filters against vectors recorded from the reference build
    (tests/golden/make_golden_tactics.py)."""
    gold = load_golden("tactics.json")
    boards = {}
    n_ladder = 0
    for pos in gold["positions"]:
        rec = gold["records"][pos["record"]]
        key = pos["record"]
        if key not in boards:
            b = OracleBoard()
            b.apply_handicap(rec["handi"])
            boards[key] = [b, 0]
        b, done = boards[key]
        for m in rec["moves"][done:pos["n_moves"]]:
            assert b.forward(m)[0]
        boards[key][1] = pos["n_moves"]
        t = b.tactics(pos["thres"])

        def dense(sp, shape, dtype):
            a = np.zeros(shape, dtype)
            for i, v in sp:
                a[i] = v
            return a
        where = "%s after %d moves" % (key, pos["n_moves"])
        assert np.array_equal(t["self_atari"], dense(pos["self_atari"], 361, np.int16)), where
        assert np.array_equal(t["gives_ko"], dense(pos["gives_ko"], 361, np.uint8)), where
        assert np.array_equal(1 - t["cand"], dense(pos["not_cand"], 361, np.uint8)), where
        assert np.array_equal(t["libs"] + 1, dense(pos["libs_plus1"], (361, 2), np.int16)), where
        assert np.array_equal(t["ladder"], dense(pos["ladder"], 361, np.int16)), where
        n_ladder += len(pos["ladder"])
    assert n_ladder >= 20
