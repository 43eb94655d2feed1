"""Host-side tests of the `go_game` module (no GPU): the pybind surface of the reference
(go/python_wrapper.cc:21-34, elf/pybind_interface.h:79-95), tensor specs (go/game_context.h:67-83),
error behaviour, and the SGF reader against what the reference's Sgf::Load produced for every
file of the test corpus (tests/golden/corpus_parse.json, made by make_golden_batch.py)."""
import os

import numpy as np
import pytest

from common import GOLDEN, load_golden


@pytest.fixture(scope="module")
def go():
    from elf_amd import go_game
    return go_game


def make_gc(go, mode="selfplay", num_games=8, **kw):
    co = go.ContextOptions()
    co.num_games = num_games
    opt = go.GameOptions()
    opt.mode = mode
    opt.seed = 3
    for k, v in kw.items():
        setattr(opt, k, v)
    return go.GameContext(co, opt), co, opt


def test_module_surface(go):
    for name in ("ContextOptions", "GameOptions", "EntryInfo", "GroupStat", "Infos", "GameContext", "TSOptions"):
        assert hasattr(go, name)
    co = go.ContextOptions()
    assert (co.num_games, co.max_num_threads, co.T, co.num_collectors) == (1, 0, 1, 1)   # python_options_utils_cpp.h:22-37
    assert not co.verbose_comm and not co.verbose_collector and not co.wait_per_group
    assert co.mcts_options.num_threads == 16 and co.mcts_options.pick_method == "most_visited"
    opt = go.GameOptions()
    assert opt.num_future_actions == 3 and opt.num_games_per_thread == 300 and opt.data_aug == -1
    assert opt.start_ratio_pre_moves == 0.5 and opt.ratio_pre_moves == 0.0 and opt.move_cutoff == -1
    gc, _, _ = make_gc(go)
    for m in ("Wait", "WaitGroup", "Steps", "Version", "PrintSummary", "CreateGroupStat", "AddCollectors", "Start",
              "Stop", "__len__", "AddTensor", "GetTensorSpec", "GetCollectorInfos", "GetParams", "ShowBoard"):
        assert hasattr(gc, m), m
    assert len(gc) == 8
    assert gc.GetParams() == dict(num_action=361, board_size=19, num_planes=25, num_future_actions=3,
                                  our_stone_plane=7, opponent_stone_plane=8)
    assert go.Infos().gid == -1 and go.Infos().batchsize() == 0
    g = go.GroupStat()
    g.hist_len, g.name = 1, "actor"
    assert g.info() == '[gid=-1][T=1][name="actor"]'


def test_tensor_specs_match_the_reference_entry_func(go):
    gc, _, _ = make_gc(go, num_future_actions=5)
    gid = gc.AddCollectors(4, 0, 0, gc.CreateGroupStat())
    want = {"s": ("float", [1, 4, 25, 19, 19]), "offline_a": ("int64_t", [1, 4, 5]), "pi": ("float", [1, 4, 361]),
            "a": ("int64_t", [1, 4]), "V": ("float", [1, 4]), "last_terminal": ("char", [1, 4]),
            "legal": ("unsigned char", [1, 4, 361])}
    for k in ("id", "seq", "game_counter", "move_idx", "winner", "aug_code", "game_record_idx"):
        want[k] = ("int32_t", [1, 4])
    for key, (typ, sz) in want.items():
        e = gc.GetTensorSpec(gid, key, 1)
        assert (e.key, e.type, list(e.sz)) == (key, typ, sz), key
    e = gc.GetTensorSpec(gid, "no_such_key", 1)       # unknown key: empty key, Python then tries last_<key>
    assert e.key == "" and list(e.sz) == [1, 4]
    assert gc.GetTensorSpec(gid, "", 1).key == ""
    assert "Collector[0] Batchsize: 4" in gc.GetCollectorInfos()


def test_errors_are_python_exceptions(go):
    gc, _, _ = make_gc(go)
    gs = gc.CreateGroupStat()
    gid = gc.AddCollectors(4, 0, 0, gs)
    buf = np.zeros((1, 4), np.int64)
    e = gc.GetTensorSpec(gid, "a", 1)
    e.p, e.byte_size = buf.ctypes.data, buf.nbytes
    with pytest.raises(Exception, match="Unknown input_reply"):     # state_collector.h:213
        gc.AddTensor(gid, "output", e)
    gc.AddTensor(gid, "reply", e)
    e.byte_size = 3
    with pytest.raises(Exception, match="bytes"):
        gc.AddTensor(gid, "reply", e)
    e.key = "bogus"
    with pytest.raises(Exception, match="unknown key"):
        gc.AddTensor(gid, "input", e)
    with pytest.raises(Exception):
        gc.GetTensorSpec(7, "s", 1)
    gs.hist_len = 4
    with pytest.raises(Exception, match="hist_len"):
        gc.AddCollectors(4, 0, 0, gs)
    with pytest.raises(Exception, match="before Start"):
        gc.Wait(0)
    with pytest.raises(Exception, match="wait_batch error"):        # state_collector.h:79-81
        gc.WaitGroup(0, 0)
    gc.Steps(go.Infos())                                            # gid -1: ignored (comm_template.h:270)
    with pytest.raises(Exception, match="use_mcts"):                # no tree search in this context
        gc.SearchInfo(0)
    co = go.ContextOptions()
    co.num_games = 2
    co.mcts_options.pick_method = "best_guess"                      # elf/tree_search.h:266-269
    opt = go.GameOptions()
    opt.mode, opt.seed, opt.use_mcts = "selfplay", 3, True
    with pytest.raises(Exception, match="Pick method unknown"):
        go.GameContext(co, opt)
    co.mcts_options.pick_method = "most_visited"
    co.mcts_options.persistent_tree = True                          # accepted since round 3
    go.GameContext(co, opt)


def test_sgf_reader_matches_the_reference_on_the_corpus(go):
    ref = load_golden("corpus_parse.json")
    assert len(ref) >= 12
    for name, want in ref.items():
        with open(os.path.join(GOLDEN, "corpus", name)) as f:
            r = go.parse_sgf(f.read())
        assert r["moves"] == want["moves"], name
        assert r["num_moves"] == want["num_moves"], name
        assert {1: 1, 2: -1}.get(r["winner"], 0) == want["winner"], name
        assert (r["handi"], r["board_size"]) == (want["handi"], want["board_size"]), name
    g = load_golden("sample1.json")                                  # the reference's own fixture
    with open(os.path.join(GOLDEN, "sample1.sgf")) as f:
        r = go.parse_sgf(f.read())
    assert r["moves"][:g["num_moves"]] == g["moves"][:g["num_moves"]] and r["num_moves"] == 258


def test_sgf_reader_quirk_texts_against_the_reference(go):
    for case in load_golden("sgf_quirks.json"):
        r = go.parse_sgf(case["text"])
        if case["ref"] is None:
            assert not r["ok"], case["text"]
            continue
        want = case["ref"]
        got = dict(moves=r["moves"], num_moves=r["num_moves"], winner={1: 1, 2: -1}.get(r["winner"], 0),
                   handi=r["handi"], board_size=r["board_size"])
        assert r["ok"] and got == want, case["text"]


def test_sgf_reader_quirks(go):
    p = go.parse_sgf
    assert not p("no header at all")["ok"]
    assert not p("(;GM[1]SZ[19])")["ok"]                            # header only: iterator done at once
    r = p("(;GM[1]SZ[19]RE[b+R]HA[4];B[pd];W[];B[tt];W[dd])")
    assert r["moves"] == [15 * 19 + 3, -1, -1, 3 * 19 + 3] and r["num_moves"] == 1
    assert r["winner"] == 1 and r["handi"] == 4
    assert p("(;GM[1];B[aa]\nW[bb];W[cc])")["moves"] == [0, 2 * 19 + 2]   # "\nW" is not a one-char key
    assert p("(;GM[1];B[aa]C[x \\] ;B[bb] y];W[cc])")["moves"] == [0, 2 * 19 + 2]   # escaped bracket in a comment
    assert p("(;GM[1];B[aa](;W[bb])(;W[cc]))")["moves"] == [0, 20, 40]       # variations are flattened
    assert p("(;GM[1];B[aa];(;W[bb]);W[cc])")["moves"] == [0]               # ";(" ends the chain
    assert p("(;GM[1];B[aa];C[no move];W[bb])")["moves"] == [0, -1, 20]
    assert p("(; SZ [19] KM[x];B[aa])")["ok"] is False                     # unparsable number


def test_aug_action_tables_match_golden(go):
    d4 = load_golden("d4.json")
    for code in range(8):
        assert [go.aug_action(code, a) for a in range(361)] == d4["coord2action"][code]
        assert [go.aug_action_inv(code, a) for a in range(361)] == d4["action2coord"][code]


def test_tar_corpus_lists_and_reads_like_the_list_file(go):
    """OfflineLoader::InitSharedBuffer takes a list file or a tar archive (go/offpolicy_loader.cc:27-50)."""
    tar = os.path.join(GOLDEN, "corpus", "corpus.tar")
    lst = os.path.join(GOLDEN, "corpus", "train.lst")
    names = go.corpus_names(lst)
    assert names == go.corpus_names(tar) == load_golden("batch_streams.json")["corpus"]
    ref = load_golden("corpus_parse.json")
    for i, nm in enumerate(names):
        a, b = go.corpus_record(tar, i), go.corpus_record(lst, i)
        assert a == b and a["moves"] == ref[nm]["moves"] and a["num_moves"] == ref[nm]["num_moves"], nm
    assert go.corpus_names(os.path.join(GOLDEN, "corpus", "missing.tar")) == []


def test_own_mersenne_twister_gives_the_standard_stream(go):
    """OfflineLoader keeps one engine per replay; its own Mt19937 (index in front of the state, so the
    two lines a draw touches can be prefetched) must give std::mt19937's stream for the seeds GoGame
    derives (go/game.cc:51), across several refills of the state."""
    for seed in (0, 1, 7, 5489, 13773 + 7, 2 ** 31, 2 ** 32 - 1, 997 * 255 * 299 + 13773 * 299 + 7 + 7):
        assert go.mt19937_matches_std(seed & 0xFFFFFFFF, 3000), seed
