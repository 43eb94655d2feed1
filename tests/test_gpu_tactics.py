"""GPU parity of goenv_tactics (heuristic move filters) against the golden vectors
recorded from the reference build, the CPU oracle and hand-built shapes.  Bit-exact."""
import numpy as np
import pytest

from common import A, load_golden
from oracle.pyoracle import OracleBoard

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

KEYS = ("self_atari", "gives_ko", "cand", "libs", "ladder")


@pytest.fixture(scope="module")
def goenv():
    from elf_amd import goenv as m
    m.load_library()
    assert torch.cuda.is_available()
    return m


def outputs(env, count):
    dev = env.device
    return dict(self_atari=torch.full((count, 361), -7, dtype=torch.int16, device=dev),
                gives_ko=torch.full((count, 361), 9, dtype=torch.uint8, device=dev),
                cand=torch.full((count, 361), 9, dtype=torch.uint8, device=dev),
                libs=torch.full((count, 361, 2), -7, dtype=torch.int16, device=dev),
                ladder=torch.full((count, 361), -7, dtype=torch.int16, device=dev))


def dense(sp, shape, dtype):
    a = np.zeros(shape, dtype)
    for i, v in sp:
        a[i] = v
    return a


def test_golden_positions_from_the_reference(goenv):
    gold = load_golden("tactics.json")
    pos = gold["positions"]
    B = len(pos)
    env = goenv.GoEnv(B, seed=1)
    stride = max(p["n_moves"] for p in pos)
    moves = np.full((B, stride), -1, np.int16)
    nm = np.zeros(B, np.int32)
    handi = np.zeros(B, np.int32)
    for b, p in enumerate(pos):
        rec = gold["records"][p["record"]]
        moves[b, :p["n_moves"]] = rec["moves"][:p["n_moves"]]
        nm[b] = p["n_moves"]
        handi[b] = rec["handi"]
    env.reset(handicap=torch.from_numpy(handi).to(env.device))
    napplied = env.new(torch.int32)
    env.replay(torch.from_numpy(moves).to(env.device), torch.from_numpy(nm).to(env.device), napplied)
    assert napplied.cpu().tolist() == nm.tolist()
    n_ladder = 0
    for thres in (1, 3, 100):
        sel = [b for b, p in enumerate(pos) if p["thres"] == thres]
        idx = torch.tensor(sel, dtype=torch.int32, device=env.device)
        out = {k: v.cpu().numpy() for k, v in env.tactics(thres, idx=idx, **outputs(env, len(sel))).items()}
        for r, b in enumerate(sel):
            p = pos[b]
            where = "%s after %d moves, thres %d" % (p["record"], p["n_moves"], thres)
            assert np.array_equal(out["self_atari"][r], dense(p["self_atari"], 361, np.int16)), where
            assert np.array_equal(out["gives_ko"][r], dense(p["gives_ko"], 361, np.uint8)), where
            assert np.array_equal(1 - out["cand"][r], dense(p["not_cand"], 361, np.uint8)), where
            assert np.array_equal(out["libs"][r] + 1, dense(p["libs_plus1"], (361, 2), np.int16)), where
            assert np.array_equal(out["ladder"][r], dense(p["ladder"], 361, np.int16)), where
            n_ladder += int((out["ladder"][r] > 0).sum())
    assert n_ladder >= 20
    env.close()


@pytest.mark.parametrize("plies,thres", [(40, 1), (150, 2), (330, 3), (420, 1)])
def test_playout_positions_match_oracle(goenv, plies, thres):
    B, seed = 96, 5 + plies
    env = goenv.GoEnv(B, seed=seed, max_ply=722)
    acts = env.new(torch.int32)
    oracles = [OracleBoard() for _ in range(B)]
    for _ in range(plies):
        env.ply(goenv.F_RANDOM_POLICY, actions_out=acts)
        for b, a in enumerate(acts.cpu().tolist()):
            oracles[b].forward(a)
    before = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    out = {k: v.cpu().numpy() for k, v in env.tactics(thres, **outputs(env, B)).items()}
    hits = 0
    for b, o in enumerate(oracles):
        want = o.tactics(thres)
        for k in KEYS:
            assert np.array_equal(out[k][b], want[k]), "board %d %s" % (b, k)
        hits += int((want["self_atari"] > 0).sum())
    assert hits > 0
    after = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    for k in before:                               # the filters leave the boards alone
        assert np.array_equal(before[k], after[k]), k
    # a contiguous range and NULL outputs
    part = env.tactics(thres, first=10, count=7, ladder=torch.zeros((7, 361), dtype=torch.int16, device=env.device))
    assert np.array_equal(part["ladder"].cpu().numpy(), out["ladder"][10:17])
    env.close()


def _stage(goenv, colours, next_player):
    """One board with the given {point: colour} stones, `next_player` to move, on GPU and oracle."""
    color = np.zeros((1, 361), np.uint8)
    for p, c in colours.items():
        color[0, p] = c
    lastp = np.where(color > 0, 1, 0).astype(np.uint16)
    sc = np.zeros((1, 16), np.int32)
    sc[0, :11] = [30, next_player, 0, 0, A(10, 10), A(11, 11), -1, -1, 5, -1, 0]
    color[0, A(10, 10)] = 3 - next_player
    color[0, A(11, 11)] = next_player
    env = goenv.GoEnv(1, seed=1)
    env.set_state(torch.from_numpy(color).to(env.device), torch.from_numpy(lastp).to(env.device),
                  torch.from_numpy(sc).to(env.device))
    o = OracleBoard()
    o.set_state(color[0], lastp[0], sc[0])
    return env, o


def test_capture_that_frees_liberties_elsewhere(goenv):
    """The one self-atari shape that is not local: the move's only liberty comes from capturing a
    multi-stone group through a single neighbour stone."""
    B_, W_ = 1, 2
    # black (0,0),(1,0) in atari at c=(2,0); white (0,1),(1,1),(2,1) in atari at c, touching the
    # black pair; white (3,0) alive.  Capturing frees (0,1),(1,1),(2,1): three liberties.
    frees = {A(0, 0): B_, A(1, 0): B_, A(0, 1): W_, A(1, 1): W_, A(2, 1): W_, A(3, 0): W_,
             A(0, 2): B_, A(1, 2): B_, A(2, 2): B_, A(3, 1): B_}
    # centre: black (5,4) in atari at c=(5,5); the white pair (5,6),(5,7) in atari at c runs away
    # from it; the other neighbours of c are living white stones.  One liberty after the capture:
    # a self-atari of two stones.
    stays = {A(5, 4): B_, A(4, 4): W_, A(6, 4): W_, A(5, 3): W_, A(4, 5): W_, A(6, 5): W_,
             A(5, 6): W_, A(5, 7): W_, A(4, 6): B_, A(6, 6): B_, A(4, 7): B_, A(6, 7): B_, A(5, 8): B_}
    for colours, c, want in ((frees, A(2, 0), 0), (stays, A(5, 5), 2)):
        env, o = _stage(goenv, colours, B_)
        out = {k: v.cpu().numpy()[0] for k, v in env.tactics(1, **outputs(env, 1)).items()}
        ref = o.tactics(1)
        for k in KEYS:
            assert np.array_equal(out[k], ref[k]), k
        assert o.check_move(c)
        assert out["self_atari"][c] == want
        env.close()


def test_argument_checks(goenv):
    env = goenv.GoEnv(4, seed=1)
    with pytest.raises(goenv.GoEnvError):
        env.tactics(-1, **outputs(env, 4))
    with pytest.raises(goenv.GoEnvError):
        env.tactics(1, first=2, count=5, ladder=torch.zeros((5, 361), dtype=torch.int16, device=env.device))
    env.tactics(1)                                  # no outputs: a no-op launch
    out = env.tactics(1, **outputs(env, 4))         # empty boards: nothing flagged, every point a candidate
    assert int(out["self_atari"].abs().sum()) == 0 and int(out["cand"].sum()) == 4 * 361
    assert int((out["libs"] != -1).sum()) == 0 and int(out["ladder"].abs().sum()) == 0
    env.close()
