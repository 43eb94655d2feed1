"""GPU soak: EVERY float of every board's planes and every legal-mask byte, on every ply, for the
launch shapes the benchmark uses (1,024 and 2,048 boards: three waves per board + write-through stores;
4,200 boards: plain stores), against oracle boards replaying the same moves.  ~250,000 board-plies:
a fault that corrupts one float in a thousand board-plies (an unguarded store-data hazard behind the
inline-asm write-through store did exactly that) cannot hide here."""
import numpy as np
import pytest

from oracle.pyoracle import OracleBoard

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("boards,plies,parts", [(1024, 120, 1), (2048, 40, 1), (1024, 60, 2), (4200, 10, 1)])
def test_all_planes_of_all_boards_every_ply(boards, plies, parts):
    from elf_amd import goenv
    goenv.load_library()
    env = goenv.GoEnv(boards, seed=17, max_ply=722)
    feat = env.new(torch.float32, 25, 19, 19)
    legal = env.new(torch.uint8, 361)
    acts = env.new(torch.int32)
    oracles = [OracleBoard() for _ in range(boards)]
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    bad = 0
    for t in range(plies):
        feat.fill_(-3.0)
        env.ply_repeat_parts(1, parts, flags, feat_out=feat, legal_out=legal, actions_out=acts)
        f = feat.cpu().numpy().reshape(boards, -1)
        lg = legal.cpu().numpy()
        for b, a in enumerate(acts.cpu().tolist()):
            o = oracles[b]
            assert o.forward(a)[0]
            if not np.array_equal(f[b], o.extract(0)) or not np.array_equal(lg[b], o.legal()):
                bad += 1
    assert bad == 0, "%d of %d board-plies differ from the oracle" % (bad, boards * plies)
    env.close()


def test_parts_give_the_same_boards_as_one_launch_sequence():
    from elf_amd import goenv
    goenv.load_library()
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    ref = None
    for parts in (1, 2, 3, 8):
        env = goenv.GoEnv(777, seed=4, max_ply=300)
        feat = env.new(torch.float32, 25, 19, 19)
        legal = env.new(torch.uint8, 361)
        env.ply_repeat_parts(650, parts, flags, feat_out=feat, legal_out=legal)
        torch.cuda.synchronize()
        st = {k: v.cpu() for k, v in env.get_state().items()}
        st["feat"], st["legal"], st["stats"] = feat.cpu(), legal.cpu(), torch.tensor(env.stats())
        env.close()
        if ref is None:
            ref = st
            assert int(ref["stats"][0]) > 0            # games finished and restarted on the way
        else:
            for k in ref:
                assert torch.equal(ref[k], st[k]), (parts, k)
    env = goenv.GoEnv(4, seed=1)
    with pytest.raises(goenv.GoEnvError):
        env.ply_repeat_parts(1, 9, flags)
    with pytest.raises(goenv.GoEnvError):
        env.ply_repeat_parts(1, 5, flags)               # more parts than boards
    env.close()


@pytest.mark.parametrize("boards,plies,launches", [(1024, 64, 2), (4100, 8, 2)])
def test_generator_rings_every_float(boards, plies, launches):
    """The bench line's launch shape (1,024 boards x 64 plies per goenv_generate launch; 4,100
    boards take the three-wave shape): every slot of the rings -- move, mask, all 9,025 floats --
    against oracle boards that replay the moves."""
    from elf_amd import goenv
    goenv.load_library()
    seed = 23
    env = goenv.GoEnv(boards, seed=seed, max_ply=722)
    dev = env.device
    feat = torch.empty(plies, boards, 25, 19, 19, device=dev)
    legal = torch.empty(plies, boards, 361, dtype=torch.uint8, device=dev)
    moves = torch.empty(plies, boards, dtype=torch.int32, device=dev)
    oracles = [OracleBoard() for _ in range(boards)]
    bad = 0
    for _ in range(launches):
        feat.fill_(-3.0)
        legal.fill_(9)
        env.generate(plies, feat, legal, moves)
        mv = moves.cpu().numpy()
        for t in range(plies):
            f = feat[t].cpu().numpy().reshape(boards, -1)
            lg = legal[t].cpu().numpy()
            for b, o in enumerate(oracles):
                assert int(mv[t, b]) == o.policy_move(seed, b, 0), (t, b)
                assert o.forward(int(mv[t, b]))[0]
                if not np.array_equal(f[b], o.extract(0)) or not np.array_equal(lg[b], o.legal()):
                    bad += 1
    assert bad == 0, "%d of %d board-plies differ from the oracle" % (bad, boards * plies * launches)
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    from common import assert_canon_equal
    for b in range(0, boards, 37):
        assert_canon_equal({k: v[b] for k, v in st.items()}, oracles[b].canonical(), "board %d" % b)
    env.close()


@pytest.mark.parametrize("boards,cap,plies,launches", [(256, 50, 64, 3), (96, 722, 64, 10)])
def test_generator_across_game_ends_against_oracle(boards, cap, plies, launches):
    """Games END inside a generator launch: the kernel scores them (Tromp-Taylor, go/board.cc:1584-1692),
    counts them into the stats vector and restarts them in place, all between two plies of one launch.
    A low ply cap ends every game several times per launch; the long run ends most games by two passes.
    Every ring slot (move, mask, all planes) and the final stats vector against oracle boards that are
    restarted where the device restarted, with wins from tt_score - komi."""
    from elf_amd import goenv
    goenv.load_library()
    seed, komi = 29, 7.5
    env = goenv.GoEnv(boards, seed=seed, komi=komi, max_ply=cap)
    dev = env.device
    feat = torch.empty(plies, boards, 25, 19, 19, device=dev)
    legal = torch.empty(plies, boards, 361, dtype=torch.uint8, device=dev)
    moves = torch.empty(plies, boards, dtype=torch.int32, device=dev)
    oracles = [OracleBoard() for _ in range(boards)]
    game = [0] * boards
    want = [0] * 16

    def ended(o):
        return o.is_game_end() or o.canonical()["scalars"][0] > cap

    def account(o):
        sc = o.canonical()["scalars"]
        n, score = int(sc[0]) - 1, o.tt_score()
        want[goenv.ST_GAMES] += 1
        margin = score - komi
        want[goenv.ST_BLACK_WINS if margin > 0 else (goenv.ST_WHITE_WINS if margin < 0 else goenv.ST_DRAWS)] += 1
        want[goenv.ST_PLIES] += n
        want[goenv.ST_BCAP] += int(sc[2])
        want[goenv.ST_WCAP] += int(sc[3])
        want[goenv.ST_SUM_LEN2] += n * n
        want[goenv.ST_PLY_CAP_HITS] += 0 if o.is_game_end() else 1
        want[goenv.ST_MAX_LEN] = max(want[goenv.ST_MAX_LEN], n)
        want[goenv.ST_SUM_SCORE] += int(score)

    bad = natural = 0
    for _ in range(launches):
        feat.fill_(-3.0)
        legal.fill_(9)
        env.generate(plies, feat, legal, moves)
        mv = moves.cpu().numpy()
        for t in range(plies):
            f = feat[t].cpu().numpy().reshape(boards, -1)
            lg = legal[t].cpu().numpy()
            for b in range(boards):
                o = oracles[b]
                if ended(o):                              # the device restarts it before this ply
                    natural += o.is_game_end()
                    account(o)
                    o = oracles[b] = OracleBoard()
                    game[b] += 1
                assert int(mv[t, b]) == o.policy_move(seed, b, game[b]), (t, b)
                assert o.forward(int(mv[t, b]))[0]
                if not np.array_equal(f[b], o.extract(0)) or not np.array_equal(lg[b], o.legal()):
                    bad += 1
    assert bad == 0, "%d board-plies differ from the oracle" % bad
    assert want[goenv.ST_GAMES] >= boards // 2
    if cap >= 700:
        assert natural > boards // 4, "too few games ended by two passes (%d)" % natural
    mid = env.stats()                                     # counted inside the launches: games already restarted
    assert mid == want, (mid, want)
    env.finish()                                          # + finished boards still waiting for their restart
    for o in oracles:
        if ended(o):
            account(o)
    assert env.stats() == want
    env.close()


CONSTANT_PLANES = [12, 13] + list(range(16, 25))


@pytest.mark.parametrize("boards,plies", [(1024, 60), (4200, 8)])
def test_constant_planes_preset_keeps_every_float(boards, plies):
    """goenv_set_constant_planes_preset: the caller zero-fills its tensors ONCE and no kernel writes planes
    12, 13, 16..24 again.  Every float of every board on every ply must still equal BoardFeature::Extract's
    (oracle), through k_ply, k_features and the generator; a sentinel written into a constant plane must
    survive (nothing writes there)."""
    from elf_amd import goenv
    goenv.load_library()
    env = goenv.GoEnv(boards, seed=33, max_ply=722)
    env.set_constant_planes_preset(True)
    feat = env.new(torch.float32, 25, 19, 19)            # zero-filled once, never again
    feat2 = env.new(torch.float32, 25, 19, 19)
    legal = env.new(torch.uint8, 361)
    acts = env.new(torch.int32)
    oracles = [OracleBoard() for _ in range(boards)]
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    bad = 0
    for t in range(plies):
        env.ply(flags, feat_out=feat, legal_out=legal, actions_out=acts)
        env.features(feat2, None, None)
        assert torch.equal(feat.view(torch.int32), feat2.view(torch.int32)), "k_features differs from k_ply at ply %d" % t
        f = feat.cpu().numpy().reshape(boards, -1)
        for b, a in enumerate(acts.cpu().tolist()):
            o = oracles[b]
            assert o.forward(a)[0]
            if not np.array_equal(f[b], o.extract(0)):
                bad += 1
    assert bad == 0, "%d of %d board-plies differ from the oracle" % (bad, boards * plies)
    feat[:, 20, 3, 3] = 7.0                               # a constant plane: nobody writes it any more
    env.ply(flags, feat_out=feat, legal_out=legal, actions_out=acts)
    assert bool((feat[:, 20, 3, 3] == 7.0).all()) and bool((feat[:, CONSTANT_PLANES].sum() == 7.0 * boards).item())
    # the generator: rings zero-filled once
    T = 4
    ring = torch.zeros(T, boards, 25, 19, 19, device=env.device)
    moves = torch.empty(T, boards, dtype=torch.int32, device=env.device)
    for b, a in enumerate(acts.cpu().tolist()):
        assert oracles[b].forward(a)[0]
    env.generate(T, ring, None, moves)
    mv = moves.cpu().numpy()
    games = np.zeros(boards, np.int64)
    for t in range(T):
        f = ring[t].cpu().numpy().reshape(boards, -1)
        for b in range(0, boards, 7):
            o = oracles[b]
            if o.is_game_end() or o.canonical()["scalars"][0] > 722:
                o = oracles[b] = OracleBoard()
            assert o.forward(int(mv[t, b]))[0]
            assert np.array_equal(f[b], o.extract(0)), (t, b)
    env.close()
