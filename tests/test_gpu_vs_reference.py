"""GPU against the REFERENCE ITSELF (north star: "checked by replaying each GPU-generated move sequence
and the SGF corpus through the reference CPU board.cc/board_feature.cc").

Every board of every launch is replayed, move by move, through the unmodified reference compiled from
/root/reference/go/{board,board_feature,go_state,sgf}.cc into oracle/_ref/libgoref.so (which travels
to the GPU box prebuilt); after every ply the reference's canonical view (go/board.h:38-125), its
FindAllValidMoves mask (go/board.cc:763-779) and all 9,025 floats of BoardFeature::Extract
(go/board_feature.cc:175-195, under the board's D4 code) are compared with what the device wrote.
No intermediary: the restatement in oracle/go_oracle.c is not loaded by this file.
Bar: bit-exact."""
import glob
import os

import numpy as np
import pytest

from common import GOLDEN, load_golden
from oracle.pyoracle import RefBatch, RefBoard

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not RefBoard.available(), reason="oracle/_ref/libgoref.so not built")]
torch = pytest.importorskip("torch")

STATE_KEYS = ("color", "liberties", "stones", "last_placed", "scalars")


@pytest.fixture(scope="module")
def goenv():
    from elf_amd import goenv as m
    m.load_library()
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return m


def host_state(env):
    return {k: v.cpu().numpy() for k, v in env.get_state().items()}


@pytest.mark.parametrize("boards,nw", [(256, 0), (1024, 0), (300, 1), (300, 2)])
def test_policy_games_to_the_end_every_board_every_ply(goenv, boards, nw):
    """cfg 2: on-device policy, one fused launch per ply, every game to its end (two passes or the ply cap).
    The reference picks the same move (its own FindAllValidMoves / IsTrueEye under the shared counter RNG),
    and after it the same state, mask and planes, for every board on every ply."""
    seed, cap = 41, 722
    env = goenv.GoEnv(boards, seed=seed, max_ply=cap)
    if nw:
        env.set_waves_per_board(nw)
    ref = RefBatch(boards)
    aug = (np.arange(boards) % 8).astype(np.int32)
    augs = torch.from_numpy(aug).to(env.device)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    acts = env.new(torch.int32)
    flags = goenv.F_RANDOM_POLICY | goenv.F_SKIP_DONE
    live = np.ones(boards, bool)
    checked = 0
    for ply in range(cap + 2):
        if not live.any():
            break
        env.ply(flags, aug_codes=augs, feat_out=feat, legal_out=legal, actions_out=acts)
        a = acts.cpu().numpy()
        want = ref.policy(seed)
        assert np.array_equal(a[live], want[live]), "ply %d: the reference's policy picks other moves" % ply
        assert (a[~live] == -3).all()
        # finished boards keep their last planes / mask: the reference is not moved and must still agree
        ref.check(np.where(live, a, -3), None, host_state(env), legal.cpu().numpy(), feat.cpu().numpy(), aug,
                  "ply %d" % ply)
        checked += int(live.sum())
        live &= ref.ended(cap) == 0
    assert not live.any()
    sc, cn, jp = env.new(torch.float32), env.new(torch.float32), env.new(torch.float32)
    env.score(sc, cn)
    env.fast_score(1, jp)
    sc, cn, jp = sc.cpu().numpy(), cn.cpu().numpy(), jp.cpu().numpy()
    for b in range(boards):
        r = ref.board(b)
        assert (sc[b], cn[b], jp[b]) == (r.tt_score(), r.fast_score(0), r.fast_score(1)), "scores of board %d" % b
    print("\n[gpu-vs-reference] %d boards, %d board-plies, every float" % (boards, checked))
    assert checked > boards * 300
    env.close()


def test_sample1_replay_all_augmentations(goenv):
    """cfg 1 on the GPU: go/sample1.sgf (read by the reference's own Sgf parser) on 8 boards, one per D4 code."""
    rec = RefBoard.sgf_load(os.path.join(GOLDEN, "sample1.sgf"))
    moves = rec["moves"][:rec["num_moves"]]
    assert len(moves) == 258
    env = goenv.GoEnv(8, seed=1)
    ref = RefBatch(8)
    aug = np.arange(8, dtype=np.int32)
    augs = torch.from_numpy(aug).to(env.device)
    feat, legal, status = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361), env.new(torch.int32)
    env.features(feat, augs, legal)
    ref.check(None, None, host_state(env), legal.cpu().numpy(), feat.cpu().numpy(), aug, "start")
    for i, m in enumerate(moves):
        env.ply(0, actions=torch.full((8,), int(m), dtype=torch.int64, device=env.device), aug_codes=augs,
                feat_out=feat, legal_out=legal, status_out=status)
        ref.check(np.full(8, int(m), np.int32), status.cpu().numpy(), host_state(env), legal.cpu().numpy(),
                  feat.cpu().numpy(), aug, "ply %d" % i)
    assert status.cpu().tolist() == [1] * 8
    sc, cn = env.new(torch.float32), env.new(torch.float32)
    env.score(sc, cn)
    assert sc.cpu().tolist() == [ref.board(0).tt_score()] * 8 == [88.0] * 8
    assert cn.cpu().tolist() == [ref.board(0).fast_score(0)] * 8 == [25.0] * 8
    env.close()


def test_sgf_corpus_replay(goenv):
    """The 13-record corpus (handicap games, an illegal move, a 9x9 record, passes): each record on a board
    of its own, handicap placed by goenv_reset and by the reference's HandicapTable, moves as the
    reference's parser reads them; legality of every move must agree (the record with the illegal move:
    both sides reject it and stay where they are), and state, mask and planes after every ply."""
    files = sorted(glob.glob(os.path.join(GOLDEN, "corpus", "*.sgf")))
    assert len(files) >= 13
    recs = [RefBoard.sgf_load(f) for f in files]
    n = len(recs)
    env = goenv.GoEnv(n, seed=2)
    ref = RefBatch(n)
    handi = np.array([r["handi"] for r in recs], np.int32)
    env.reset(handicap=torch.from_numpy(handi).to(env.device))
    for b, r in enumerate(recs):
        ref.board(b).apply_handicap(int(r["handi"]))
    rng = np.random.RandomState(3)
    aug = rng.randint(0, 8, n).astype(np.int32)
    augs = torch.from_numpy(aug).to(env.device)
    feat, legal, status = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361), env.new(torch.int32)
    env.features(feat, augs, legal)
    ref.check(None, None, host_state(env), legal.cpu().numpy(), feat.cpu().numpy(), aug, "after handicap")
    T = max(len(r["moves"]) for r in recs)
    illegal = 0
    for t in range(T):
        a = np.array([int(r["moves"][t]) if t < len(r["moves"]) else -1 for r in recs], np.int32)
        env.ply(0, actions=torch.from_numpy(a.astype(np.int64)).to(env.device), aug_codes=augs, feat_out=feat,
                legal_out=legal, status_out=status)
        st = status.cpu().numpy()
        illegal += int((st == 0).sum())
        ref.check(a, st, host_state(env), legal.cpu().numpy(), feat.cpu().numpy(), aug, "corpus ply %d" % t)
    assert illegal > 0, "the corpus holds a record with an illegal move"
    env.close()


def test_fused_selfplay_step_against_the_reference(goenv):
    """What GameContext.Steps issues in selfplay (goenv_ply_policy_sel: move sampled from a reply pi, played,
    next planes written, one launch; two launches above 3,072 boards).  The sampler is this build's own
    (no counterpart in the reference), so the reference replays the moves the device reports; everything after the
    choice -- legality, state, mask, planes -- is the reference's."""
    for boards, plies in ((1024, 200), (4096, 24)):
        env = goenv.GoEnv(boards, seed=9, max_ply=722)
        ref = RefBatch(boards)
        feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
        acts, status = env.new(torch.int32), env.new(torch.int32)
        g = torch.Generator(device=env.device)
        g.manual_seed(5)
        zero = np.zeros(boards, np.int32)
        for t in range(plies):
            pi = torch.rand(boards, 361, device=env.device, generator=g)
            env.ply_policy_sel(pi, feat_out=feat, legal_out=legal, actions_out=acts, status_out=status)
            assert status.cpu().tolist() == [1] * boards
            ref.check(acts.cpu().numpy(), None, host_state(env), legal.cpu().numpy(), feat.cpu().numpy(), zero,
                      "%d boards, step %d" % (boards, t))
        env.close()


@pytest.mark.parametrize("boards,plies,launches,cap,kernel,with_aug", [
    (1024, 64, 3, 90, "k_generate_pipe", False),          # the bench line's shape
    (2600, 12, 4, 40, "k_generate_pipe_multi", True),     # several boards per workgroup, D4 codes, ragged last turn
])
def test_generator_rings_against_the_reference(goenv, boards, plies, launches, cap, kernel, with_aug):
    """goenv_generate through both pipelined kernels: every ring slot -- move, mask, all planes -- against reference boards
    that are restarted where the device restarts its games."""
    seed = 31
    env = goenv.GoEnv(boards, seed=seed, max_ply=cap)
    dev = env.device
    ref = RefBatch(boards)
    feat = torch.empty(plies, boards, 25, 19, 19, device=dev)
    legal = torch.empty(plies, boards, 361, dtype=torch.uint8, device=dev)
    moves = torch.empty(plies, boards, dtype=torch.int32, device=dev)
    games = np.zeros(boards, np.int32)
    zero = ((np.arange(boards) * 3) % 8).astype(np.int32) if with_aug else np.zeros(boards, np.int32)
    augs = torch.from_numpy(zero).to(dev) if with_aug else None
    restarts = 0
    for _ in range(launches):
        feat.fill_(-3.0)
        legal.fill_(9)
        env.generate(plies, feat, legal, moves, aug_codes=augs)
        assert env.generate_kernel() == kernel
        mv = moves.cpu().numpy()
        for t in range(plies):
            for b in np.flatnonzero(ref.ended(cap)):      # the device restarts these before this ply
                ref.board(int(b)).clear()
                games[b] += 1
                restarts += 1
            assert np.array_equal(mv[t], ref.policy(seed, games)), "slot %d: moves" % t
            ref.check(mv[t], None, None, legal[t].cpu().numpy(), feat[t].cpu().numpy(), zero, "slot %d" % t)
    assert restarts >= boards
    ref.check(None, None, host_state(env), None, None, None, "records after the last launch")
    env.close()


def test_move_filters_against_the_reference(goenv):
    """SURVEY 8f-3 on the GPU box against the reference's own IsSelfAtari / IsMoveGivingSimpleKo /
    FindAllCandidateMoves / find_only_liberty / find_two_liberties / CheckLadder (go/board.cc:168-391,685-715,
    1042-1081; goref_tactics): goenv_tactics for all 361 points of 192 boards at several stages of policy games."""
    boards, seed = 192, 57
    env = goenv.GoEnv(boards, seed=seed, max_ply=722)
    ref = RefBatch(boards)
    acts = env.new(torch.int32)
    dev = env.device
    hits = dict(self_atari=0, gives_ko=0, ladder=0)
    ply = 0
    for stop, thres in ((30, 1), (90, 2), (170, 3), (260, 1), (340, 5)):
        while ply < stop:
            env.ply(goenv.F_RANDOM_POLICY, actions_out=acts)
            ref.check(acts.cpu().numpy(), None, None, None, None, None, "ply %d" % ply)
            ply += 1
        out = dict(self_atari=torch.full((boards, 361), -7, dtype=torch.int16, device=dev),
                   gives_ko=torch.full((boards, 361), 9, dtype=torch.uint8, device=dev),
                   cand=torch.full((boards, 361), 9, dtype=torch.uint8, device=dev),
                   libs=torch.full((boards, 361, 2), -7, dtype=torch.int16, device=dev),
                   ladder=torch.full((boards, 361), -7, dtype=torch.int16, device=dev))
        got = {k: v.cpu().numpy() for k, v in env.tactics(thres, **out).items()}
        for b in range(boards):
            want = ref.board(b).tactics(thres)
            for k in want:
                assert np.array_equal(got[k][b], want[k]), "ply %d board %d: %s" % (ply, b, k)
            hits["self_atari"] += int((want["self_atari"] > 0).sum())
            hits["gives_ko"] += int(want["gives_ko"].sum())
            hits["ladder"] += int((want["ladder"] > 0).sum())
    assert hits["self_atari"] > 1000 and hits["gives_ko"] > 20 and hits["ladder"] > 20, hits
    env.close()
