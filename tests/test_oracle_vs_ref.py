"""CPU: differential fuzz of the plain-C oracle against the unmodified reference build
(oracle/_ref/libgoref.so).  Skipped where the reference build is absent."""
import numpy as np
import pytest

from common import assert_canon_equal, assert_planes_equal
from oracle.pyoracle import OracleBoard, RefBoard

pytestmark = pytest.mark.skipif(not RefBoard.available(), reason="oracle/_ref/libgoref.so not built")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_games_with_passes_illegal_probes_and_handicap(seed):
    rng = np.random.default_rng(seed)
    for g in range(6):
        r, o = RefBoard(), OracleBoard()
        if g % 3 == 0:
            h = int(rng.integers(0, 10))
            r.apply_handicap(h)
            o.apply_handicap(h)
        fill_eyes = g % 2 == 0
        for ply in range(650):
            m = r.legal() if fill_eyes else r.candidates()
            idx = np.flatnonzero(m)
            u = rng.random()
            if u < 0.02 or len(idx) == 0:
                a = -1
            elif u < 0.07:
                a = int(rng.integers(0, 361))
            else:
                a = int(rng.choice(idx))
            (lr, er), (lo, eo) = r.forward(a), o.forward(a)
            assert (lr, er) == (lo, eo), (g, ply, a)
            assert_canon_equal(r.canonical(), o.canonical(), "game %d ply %d" % (g, ply))
            assert np.array_equal(r.legal(), o.legal()) and np.array_equal(r.candidates(), o.candidates())
            if ply % 5 == 0:
                aug = int(rng.integers(0, 8))
                assert_planes_equal(r.extract(aug), o.extract(aug), "game %d ply %d aug %d" % (g, ply, aug))
                assert r.tt_score() == o.tt_score()
                assert r.fast_score(0) == o.fast_score(0) and r.fast_score(1) == o.fast_score(1)
            if er and ply > 5:
                break


def test_native_differential_fuzz_two_million_plies(capsys):
    """oracle/diff_fuzz.cc: reference board.cc / board_feature.cc against the restatement inside one C
    loop ("millions of random plies").  Every ply: forward() result, game-end flag, canonical
    view, legal and candidate masks; on shares of the plies all 9,025 plane floats under a random D4 code,
    Tromp-Taylor and both fast scores, TryPlay2 of every point, the heuristic filters, the policy move."""
    import os
    from oracle import pyoracle
    if not os.path.exists(pyoracle.DIFF_SO):
        pytest.skip("oracle/_ref/libgodiff.so not built")
    target = int(os.environ.get("GOENV_FUZZ_PLIES", "2000000"))
    err, n = pyoracle.diff_fuzz(20261111, target, min(8, os.cpu_count() or 1))
    with capsys.disabled():
        print("\n[diff_fuzz] " + " ".join("%s=%d" % kv for kv in n.items()))
    assert err is None, err
    assert n["plies"] >= target
    # the fuzz reaches what it is meant to reach
    assert n["illegal"] > target // 100 and n["passes"] > target // 200 and n["ko_set"] > target // 1000
    assert n["captures"] > target // 50 and n["extracts"] > target // 8 and n["scores"] > target // 16
    assert n["handicap_games"] > 100 and n["game_ends"] > 100 and n["resigns"] > 10 and n["tactics"] > 1000
    assert n["max_groups"] < 165


def test_policy_and_d4_agree():
    for bid in range(4):
        r, o = RefBoard(), OracleBoard()
        for _ in range(120):
            a = r.policy_move(99, bid, 3)
            assert a == o.policy_move(99, bid, 3)
            r.forward(a)
            o.forward(a)
    for code in range(8):
        for a in range(361):
            assert RefBoard.aug_action(code, a) == OracleBoard.aug_action(code, a)
            assert RefBoard.aug_action_inv(code, a) == OracleBoard.aug_action_inv(code, a)


def _tactics_equal(r, o, thres, where):
    a, b = r.tactics(thres), o.tactics(thres)
    for k in a:
        assert np.array_equal(a[k], b[k]), (where, k)
    return a


def test_move_filters_on_game_records_and_playouts():
    """IsSelfAtari / IsMoveGivingSimpleKo / FindAllCandidateMoves / find_*_libert* / CheckLadder
    (SURVEY 8f-3): the restatement against the reference on every second position of the corpus
    records (real games: ladders up to depth 45) and on random playouts."""
    import glob
    import os
    corpus = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "corpus")
    hits = dict(sa=0, ko=0, ladder=0)
    for f in sorted(glob.glob(os.path.join(corpus, "*.sgf"))):
        rec = RefBoard.sgf_load(f)
        r, o = RefBoard(), OracleBoard()
        r.apply_handicap(rec["handi"])
        o.apply_handicap(rec["handi"])
        for i, m in enumerate(rec["moves"]):
            ok, _ = r.forward(int(m))
            o.forward(int(m))
            if not ok:
                break
            if i % 2 == 0:
                a = _tactics_equal(r, o, [1, 3, 100][i % 3], (f, i))
                hits["sa"] += int((a["self_atari"] > 0).sum())
                hits["ko"] += int(a["gives_ko"].sum())
                hits["ladder"] += int((a["ladder"] > 0).sum())
    assert hits["sa"] > 1000 and hits["ko"] > 50 and hits["ladder"] > 50, hits
    rng = np.random.default_rng(5)
    for g in range(4):
        r, o = RefBoard(), OracleBoard()
        for ply in range(500):
            idx = np.flatnonzero(r.candidates())
            a = -1 if len(idx) == 0 else int(rng.choice(idx))
            _, end = r.forward(a)
            o.forward(a)
            if ply % 4 == 0:
                _tactics_equal(r, o, [1, 2, 5][ply % 3], ("playout", g, ply))
            if end:
                break
