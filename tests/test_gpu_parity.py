"""GPU parity tests: the HIP path (through the C-ABI of libgoenv.so) against the CPU oracle on
the same inputs, and against the committed golden vectors of the unmodified reference.
Bar: bit-exact (integer state, u8 masks, float32 bit patterns)."""
import os

import numpy as np
import pytest

from common import (EDGE_CASES, GOLDEN, assert_canon_equal, assert_planes_equal, load_golden, position_digest)
from oracle.pyoracle import OracleBoard

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("waves_per_board")]

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def goenv():
    from elf_amd import goenv as m
    m.load_library()  # raises if the HIP library is missing: no fallback
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return m


def teq(a, b):
    """torch.equal that also works for uint16 tensors (compared through an int16 view)."""
    if a.dtype == torch.uint16:
        a, b = a.view(torch.int16), b.view(torch.int16)
    return torch.equal(a, b)


def check_boards(env, oracles, augs, feat, legal, where):
    """Compare every board's canonical view, legal mask and planes with its oracle."""
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    f = feat.cpu().numpy().reshape(env.n, -1)
    lg = legal.cpu().numpy()
    for b, o in enumerate(oracles):
        w = "%s board %d" % (where, b)
        assert_canon_equal({k: v[b] for k, v in st.items()}, o.canonical(), w)
        assert np.array_equal(lg[b], o.legal()), w + " legal mask"
        assert_planes_equal(f[b], o.extract(int(augs[b])), w)
    return st, f, lg


def test_sample1_replay_all_augmentations(goenv):
    """cfg 1 on the GPU: replay go/sample1.sgf on 8 boards, one per D4 code; every ply is checked
    against the oracle and the reference's golden digests."""
    g = load_golden("sample1.json")
    env = goenv.GoEnv(8, seed=1)
    dev = env.device
    augs = torch.arange(8, dtype=torch.int32, device=dev)
    feat = env.new(torch.float32, 25, 19, 19)
    legal = env.new(torch.uint8, 361)
    status = env.new(torch.int32)
    oracles = [OracleBoard() for _ in range(8)]
    env.features(feat, augs, legal)
    st, f, lg = check_boards(env, oracles, range(8), feat, legal, "start")
    for i in range(g["num_moves"]):
        a = g["moves"][i]
        actions = torch.full((8,), a, dtype=torch.int64, device=dev)
        # fused: step + legal + planes in one launch
        env.ply(0, actions=actions, aug_codes=augs, feat_out=feat, legal_out=legal, status_out=status)
        for o in oracles:
            assert o.forward(a)[0]
        assert status.cpu().tolist() == [1] * 8
        st, f, lg = check_boards(env, oracles, range(8), feat, legal, "ply %d" % i)
        d = position_digest({k: v[0] for k, v in st.items()}, lg[0], f[0])
        assert d == g["digests"][i + 1], "golden digest, ply %d" % i
    sc, fs = env.new(torch.float32), env.new(torch.float32)
    env.score(sc, fs)
    assert sc.cpu().tolist() == [88.0] * 8 and fs.cpu().tolist() == [25.0] * 8
    d4 = load_golden("d4.json")
    for code in range(8):
        d = position_digest({k: v[code] for k, v in st.items()}, lg[code], f[code])
        assert d == d4["sample1_final_digests"][code]
    env.close()


def test_random_policy_playouts_ply_by_ply(goenv):
    """cfg 2 at small size: on-device policy, one fused launch per ply; every board's move
    sequence is replayed through the oracle: same move chosen, same state, mask and planes."""
    g = load_golden("playouts.json")
    B = 24
    env = goenv.GoEnv(B, seed=g["seed"], max_ply=g["max_ply"])
    dev = env.device
    augs = (torch.arange(B, dtype=torch.int32, device=dev) % 8).contiguous()
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    acts, status = env.new(torch.int32), env.new(torch.int32)
    oracles = [OracleBoard() for _ in range(B)]
    live = [True] * B
    seqs = [[] for _ in range(B)]
    flags = goenv.F_RANDOM_POLICY | goenv.F_SKIP_DONE
    for ply in range(g["max_ply"] + 2):
        if not any(live):
            break
        env.ply(flags, aug_codes=augs, feat_out=feat, legal_out=legal, actions_out=acts, status_out=status)
        a_host = acts.cpu().tolist()
        for b in range(B):
            if not live[b]:
                assert a_host[b] == -3
                continue
            o = oracles[b]
            want = o.policy_move(g["seed"], b, 0)
            assert a_host[b] == want, "board %d ply %d: GPU chose %d, oracle %d" % (b, ply, a_host[b], want)
            ok, end = o.forward(want)
            assert ok
            seqs[b].append(want)
            if end or o.canonical()["scalars"][0] > g["max_ply"]:
                live[b] = False
        if ply % 3 == 0 or not any(live):
            check_boards(env, oracles, augs.cpu().tolist(), feat, legal, "ply %d" % ply)
    assert not any(live)
    st = env.get_state()
    done = st["scalars"][:, goenv.S_DONE].cpu().tolist()
    assert all(d in (1, 2) for d in done)
    # golden: the first boards' sequences are the reference's own
    for game in g["games"]:
        assert seqs[game["board_id"]] == game["moves"]
    sc, fs = env.new(torch.float32), env.new(torch.float32)
    env.score(sc, fs)
    jp = env.new(torch.float32)
    env.fast_score(1, jp)
    for b, o in enumerate(oracles):
        assert sc[b].item() == o.tt_score() and fs[b].item() == o.fast_score(0) and jp[b].item() == o.fast_score(1)
    # end-of-episode statistics
    env.finish()
    stats = env.stats()
    assert stats[goenv.ST_GAMES] == B
    assert stats[goenv.ST_PLIES] == sum(len(s) for s in seqs)
    assert stats[goenv.ST_MAX_LEN] == max(len(s) for s in seqs)
    assert stats[goenv.ST_BLACK_WINS] == sum(1 for o in oracles if o.tt_score() - 7.5 > 0)
    assert stats[goenv.ST_WHITE_WINS] == B - stats[goenv.ST_BLACK_WINS]
    env.finish()  # idempotent: finished boards are counted once
    assert env.stats() == stats
    env.close()


def test_persistent_playout_matches_per_ply_path(goenv):
    g = load_golden("playouts.json")
    B = 64
    env = goenv.GoEnv(B, seed=g["seed"], max_ply=g["max_ply"])
    moves = torch.zeros((B, 768), dtype=torch.int16, device=env.device)
    nm = env.new(torch.int32)
    env.playout(moves, nm)
    mv, n = moves.cpu().numpy(), nm.cpu().numpy()
    for game in g["games"]:
        b = game["board_id"]
        assert mv[b, :n[b]].tolist() == game["moves"]
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    for b in range(B):
        o = OracleBoard()
        seq = o.playout(g["seed"], b, 0, g["max_ply"])
        assert mv[b, :n[b]].tolist() == seq.tolist(), "board %d" % b
        assert_canon_equal({k: v[b] for k, v in st.items()}, o.canonical(), "board %d" % b)
    env.close()


def test_edge_cases_and_illegal_moves(goenv):
    names = sorted(EDGE_CASES)
    gold = load_golden("edge.json")
    B = len(names)
    env = goenv.GoEnv(B, seed=3)
    dev = env.device
    feat, legal, status = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361), env.new(torch.int32)
    sc, fs = env.new(torch.float32), env.new(torch.float32)
    oracles = [OracleBoard() for _ in range(B)]
    T = max(len(EDGE_CASES[n]) for n in names)
    for t in range(T):
        # boards whose script is over pass (legal, harmless)
        acts = [EDGE_CASES[n][t] if t < len(EDGE_CASES[n]) else -1 for n in names]
        env.step(torch.tensor(acts, dtype=torch.int64, device=dev), legal_out=legal, status_out=status)
        env.features(feat)
        env.score(sc, fs)
        stat = status.cpu().tolist()
        for b, n in enumerate(names):
            ok, _ = oracles[b].forward(acts[b])
            assert stat[b] == int(ok), (n, t)
            if t < len(EDGE_CASES[n]):
                assert stat[b] == gold[n]["legal"][t], (n, t)
        st, f, lg = check_boards(env, oracles, [0] * B, feat, legal, "edge t=%d" % t)
        for b, n in enumerate(names):
            if t < len(EDGE_CASES[n]):
                d = position_digest({k: v[b] for k, v in st.items()}, lg[b], f[b])
                assert d == gold[n]["digests"][t], (n, t)
                assert sc[b].item() == gold[n]["tt_score"][t] and fs[b].item() == gold[n]["fast_score_cn"][t]
    env.close()


def test_handicap_reset_and_partial_reset(goenv):
    gold = load_golden("handicap.json")
    env = goenv.GoEnv(11, seed=5)
    dev = env.device
    env.reset(handicap=torch.arange(11, dtype=torch.int32, device=dev))
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    oracles = []
    for h in range(11):
        o = OracleBoard()
        o.apply_handicap(h)
        oracles.append(o)
    env.features(feat, None, legal)
    st, f, lg = check_boards(env, oracles, [0] * 11, feat, legal, "handicap")
    for h in range(11):
        assert [int(p) for p in np.flatnonzero(st["color"][h] == 1)] == gold[str(h)]["stones"]
        assert st["scalars"][h][:11].tolist() == gold[str(h)]["scalars"][:11]
    for step in range(3):
        a = gold["0"]["after"][step]["action"]
        env.step(torch.full((11,), a, dtype=torch.int64, device=dev), legal_out=legal)
        env.features(feat)
        for o in oracles:
            o.forward(a)
        st, f, lg = check_boards(env, oracles, [0] * 11, feat, legal, "handicap step %d" % step)
        for h in range(11):
            d = position_digest({k: v[h] for k, v in st.items()}, lg[h], f[h])
            assert d == gold[str(h)]["after"][step]["digest"]
    # partial reset: only boards 1 and 4, board 4 with 3 stones
    env.reset(idx=torch.tensor([1, 4], dtype=torch.int32, device=dev),
              handicap=torch.tensor([0, 3], dtype=torch.int32, device=dev))
    oracles[1] = OracleBoard()
    oracles[4] = OracleBoard()
    oracles[4].apply_handicap(3)
    env.features(feat, None, legal)
    check_boards(env, oracles, [0] * 11, feat, legal, "partial reset")
    env.close()


def test_set_state_roundtrip_and_full_arrays(goenv):
    """Stage the reference's dumped positions with set_state; groups/liberties are rebuilt on
    device and must reproduce the dumped liberties, stones, mask and planes."""
    z = np.load(os.path.join(GOLDEN, "planes_full.npz"))
    names = ["sample1_final", "playout0", "playout1", "playout2"]
    env = goenv.GoEnv(len(names), seed=7)
    dev = env.device
    color = torch.tensor(np.stack([z[n + "_color"] for n in names]), device=dev)
    lastp = torch.from_numpy(np.stack([z[n + "_last_placed"] for n in names]).astype(np.uint16)).to(dev)
    scal = torch.tensor(np.stack([z[n + "_scalars"] for n in names]), device=dev)
    env.set_state(color, lastp, scal)
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    augs = torch.tensor([0, 0, 1, 2], dtype=torch.int32, device=dev)
    env.features(feat, augs, legal)
    for b, n in enumerate(names):
        ref = {k: z["%s_%s" % (n, k)] for k in ("color", "liberties", "stones", "last_placed", "scalars")}
        assert_canon_equal({k: v[b] for k, v in st.items()}, ref, n)
        assert np.array_equal(legal[b].cpu().numpy(), z[n + "_legal"])
        planes = z[n + "_planes_aug0"] if n == "sample1_final" else z[n + "_planes"]
        assert_planes_equal(feat[b].cpu().numpy(), planes, n)
    env.close()


def test_argument_errors_are_reported_not_fatal(goenv):
    env = goenv.GoEnv(4)
    with pytest.raises(goenv.GoEnvError):
        env.ply(0)  # no actions and no random policy
    with pytest.raises(goenv.GoEnvError):
        env.step(torch.zeros(3, dtype=torch.int64, device=env.device))  # wrong batch
    with pytest.raises(goenv.GoEnvError):
        goenv.GoEnv(0)
    env.close()


@pytest.mark.parametrize("B", [1024])
def test_full_size_properties(goenv, B):
    """cfg 2 at BASELINE size (1,024 boards): size-independent properties plus an oracle replay of
    a sample of boards.  (a) the persistent kernel and the per-ply path end in identical states;
    (b) every final position's stored liberties equal a from-scratch rebuild (set_state of its own
    colours reproduces the record's canonical view); (c) stats sum over boards; (d) legal mask of
    features() equals the mask of the fused ply."""
    seed = 1
    env = goenv.GoEnv(B, seed=seed, max_ply=722)
    moves = torch.zeros((B, 768), dtype=torch.int16, device=env.device)
    nm = env.new(torch.int32)
    env.playout(moves, nm)
    st_a = {k: v.clone() for k, v in env.get_state().items()}

    env2 = goenv.GoEnv(B, seed=seed, max_ply=722)
    feat, legal, legal2 = env2.new(torch.float32, 25, 19, 19), env2.new(torch.uint8, 361), env2.new(torch.uint8, 361)
    flags = goenv.F_RANDOM_POLICY | goenv.F_SKIP_DONE
    nply = int(nm.max().item())
    for _ in range(nply):
        env2.ply(flags, feat_out=feat, legal_out=legal)
    st_b = env2.get_state()
    for k in st_a:
        assert teq(st_a[k], st_b[k]), k
    env2.features(None, None, legal2)
    # boards that were still live on the last ply carry that ply's mask
    last = (nm == nply)
    assert torch.equal(legal[last], legal2[last])

    # (b) rebuild from colours
    env3 = goenv.GoEnv(B, seed=seed)
    env3.set_state(st_a["color"], st_a["last_placed"], st_a["scalars"])
    st_c = env3.get_state()
    for k in ("color", "liberties", "stones", "last_placed"):
        assert teq(st_a[k], st_c[k]), k

    # (c) stats
    env.finish()
    stats = env.stats()
    assert stats[goenv.ST_GAMES] == B and stats[goenv.ST_PLIES] == int(nm.sum().item())
    assert stats[goenv.ST_BCAP] == int(st_a["scalars"][:, goenv.S_BCAP].sum().item())
    assert stats[goenv.ST_WCAP] == int(st_a["scalars"][:, goenv.S_WCAP].sum().item())

    # oracle replay of a sample
    mv, n = moves.cpu().numpy(), nm.cpu().numpy()
    sta = {k: v.cpu().numpy() for k, v in st_a.items()}
    for b in list(range(0, B, 97)) + [B - 1]:
        o = OracleBoard()
        seq = o.playout(seed, b, 0, 722)
        assert mv[b, :n[b]].tolist() == seq.tolist(), "board %d" % b
        assert_canon_equal({k: v[b] for k, v in sta.items()}, o.canonical(), "board %d" % b)
    for e in (env, env2, env3):
        e.close()
