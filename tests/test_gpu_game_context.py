"""GPU tests of the GameContext boundary (elf_amd/go_game, C++/pybind11 over the C-ABI).

* offline mode (cfg 3): the per-game sample streams -- labels, future moves and the 25 planes --
  must equal, bit for bit, what the UNMODIFIED reference go_game module emitted for the same
  GameOptions over the same SGF corpus (tests/golden/batch_streams.json).
* selfplay / online mode: every batch row is checked against the CPU oracle replaying the same
  actions; sequencing invariants of check.py:33-91.
Host tensors (numpy, as the reference's callers pass) and device tensors (torch-ROCm) are both
exercised."""
import os

import numpy as np
import pytest

from common import GOLDEN, assert_planes_equal, load_golden
from oracle.pyoracle import FNV_SEED, OracleBoard, fnv_planes

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("waves_per_board")]
torch = pytest.importorskip("torch")

NP_TYPES = {"int32_t": "i4", "int64_t": "i8", "float": "f4", "unsigned char": "u1", "char": "i1"}
TORCH_TYPES = {"int32_t": torch.int32, "int64_t": torch.int64, "float": torch.float32, "unsigned char": torch.uint8,
               "char": torch.int8}
LABELS = ["id", "seq", "game_counter", "last_terminal", "move_idx", "winner", "aug_code", "game_record_idx"]


@pytest.fixture(scope="module")
def go():
    from elf_amd import go_game
    assert torch.cuda.is_available()
    return go_game


def register(gc, gid, which, keys, device):
    bufs = {}
    for key in keys:
        info = gc.GetTensorSpec(gid, key, 1)
        assert info.key == key
        if device:
            v = torch.zeros(*info.sz, dtype=TORCH_TYPES[info.type], device="cuda:0")
            info.p, info.byte_size = v.data_ptr(), v.numel() * v.element_size()
        else:
            v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
            info.p, info.byte_size = v.ctypes.data, v.nbytes
        gc.AddTensor(gid, which, info)
        bufs[key] = v
    return bufs


def host(v):
    return v.cpu().numpy() if isinstance(v, torch.Tensor) else v


@pytest.mark.parametrize("device", [False, True])
def test_offline_streams_equal_the_reference_module(go, device):
    check_offline_streams(go, device, "batch_streams.json")


@pytest.mark.parametrize("device", [False, True])
def test_cfg3_batch_256_equals_the_reference_module(go, device):
    """BASELINE.json configs[2] at its stated size: 256 games, ONE batch of 256 per Wait, random D4, k = 3
    (tests/golden/batch256_streams.json, made by the compiled reference module with the same options)."""
    check_offline_streams(go, device, "batch256_streams.json", full_batches=True)


def check_offline_streams(go, device, golden, full_batches=False):
    gold = load_golden(golden)
    for run in gold["runs"]:
        cfg = run["cfg"]
        co = go.ContextOptions()
        co.num_games = cfg["num_games"]
        opt = go.GameOptions()
        opt.seed, opt.mode = cfg["seed"], "offline"
        opt.list_filename = os.path.join(GOLDEN, "corpus", cfg["lst"])
        opt.data_aug, opt.start_ratio_pre_moves, opt.ratio_pre_moves = cfg["data_aug"], cfg["start_ratio"], cfg["ratio"]
        opt.move_cutoff, opt.num_games_per_thread, opt.num_future_actions = cfg["cutoff"], cfg["games_per_thread"], cfg["k"]
        gc = go.GameContext(co, opt)
        gs = gc.CreateGroupStat()
        B = cfg["num_games"]
        gid = gc.AddCollectors(B, 0, 0, gs)
        bufs = register(gc, gid, "input", ["s", "offline_a"] + LABELS, device)
        gc.Start()
        n_check = min(cfg["samples"], 150 if device else cfg["samples"])
        streams = {g: [] for g in range(B)}
        while min(len(v) for v in streams.values()) < n_check:
            infos = gc.Wait(0)
            assert infos.gid == gid and len(infos.s) == B
            h = {k: host(v) for k, v in bufs.items()}
            if full_batches:                                   # every game exactly once per batch
                assert sorted(int(g) for g in h["id"][0]) == list(range(B))
            for r in range(len(infos.s)):
                g = int(h["id"][0, r])
                assert g == infos.s[r]
                streams[g].append([int(h[k][0, r]) for k in LABELS] + [[int(a) for a in h["offline_a"][0, r]]] +
                                  ["%016x" % fnv_planes(FNV_SEED, h["s"][0, r].reshape(-1))])
            gc.Steps(infos)
        # the host never asks the device whether a record move is legal (it knows each record's legal prefix):
        # the device, which checks every move all the same, must never have disagreed
        assert gc.OfflineBadMoves() == 0
        gc.Stop()
        for g in range(B):
            want = run["streams"][str(g)][:n_check]
            got = streams[g][:n_check]
            for i, (a, b) in enumerate(zip(got, want)):
                assert a == b, "run %s game %d sample %d: got %s, reference %s" % (cfg["name"], g, i, a, b)


@pytest.mark.parametrize("device", [False, True])
def test_selfplay_loop_against_oracle(go, device):
    """df_selfplay shape: `actor` collector, inputs s (+legal, labels), reply a.  The callback
    plays a deterministic legal move (or sometimes an illegal one: the game restarts, as
    go/game.cc:81-90).  Every row's planes/mask/move_idx are checked against an oracle board
    replaying the same actions."""
    N, B = 12, 4
    co = go.ContextOptions()
    co.num_games = N
    opt = go.GameOptions()
    opt.mode, opt.seed = "selfplay", 5
    gc = go.GameContext(co, opt)
    gs = gc.CreateGroupStat()
    gs.name = "actor"
    gids = [gc.AddCollectors(B, 0, 0, gs) for _ in range(2)]
    ins = [register(gc, gid, "input", ["s", "legal"] + LABELS, device) for gid in gids]
    outs = [register(gc, gid, "reply", ["a", "pi", "V"], device) for gid in gids]
    gc.Start()
    oracles = [OracleBoard() for _ in range(N)]
    seq = [0] * N
    rng = np.random.RandomState(0)
    restarts = 0
    for it in range(120):
        infos = gc.Wait(0)
        assert infos.gid in gids
        k = gids.index(infos.gid)
        h = {key: host(v) for key, v in ins[k].items()}
        ids = list(infos.s)
        assert len(set(ids)) == len(ids) == B                      # check.py: ids unique within a batch
        acts = []
        for r, g in enumerate(ids):
            o = oracles[g]
            assert h["id"][0, r] == g
            assert h["seq"][0, r] == seq[g]                          # seq consecutive per game (run-on mode)
            assert h["game_counter"][0, r] == 0 and h["last_terminal"][0, r] == 0
            seq[g] += 1
            assert h["move_idx"][0, r] == o.canonical()["scalars"][0]
            assert h["winner"][0, r] == 0 and h["game_record_idx"][0, r] == -1 and h["aug_code"][0, r] == 0
            assert np.array_equal(h["legal"][0, r], o.legal())
            assert_planes_equal(h["s"][0, r], o.extract(0), "iter %d game %d" % (it, g))
            legal = np.flatnonzero(o.legal())
            if rng.rand() < 0.05 or legal.size == 0:
                stones = np.flatnonzero(o.canonical()["color"])
                a = int(stones[0]) if stones.size and rng.rand() < 0.7 else -1   # occupied point: illegal
            else:
                a = int(legal[rng.randint(legal.size)])
            acts.append(a)
            ok, _ = o.forward(a)
            if not ok:
                oracles[g] = OracleBoard()
                restarts += 1
        a_t = outs[k]["a"]
        if device:
            a_t.copy_(torch.tensor([acts], dtype=torch.int64))
        else:
            a_t[0, :] = acts
        gc.Steps(infos)
    assert restarts > 0
    assert "X" in gc.ShowBoard(0) or "O" in gc.ShowBoard(0)
    gc.Stop()


def test_selfplay_policy_sampling_and_episode_signals(go):
    """Reply carries only pi/V (the reference's selfplay descriptor, go/game.py:79-87): the move is
    sampled on the device from pi over the legal points.  With signal_restart + end_on_two_passes
    finished games are scored, counted and restarted with game_counter/last_terminal set."""
    N = B = 16
    co = go.ContextOptions()
    co.num_games = N
    opt = go.GameOptions()
    opt.mode, opt.seed, opt.signal_restart, opt.end_on_two_passes, opt.ply_cap = "selfplay", 9, True, True, 60
    gc = go.GameContext(co, opt)
    gs = gc.CreateGroupStat()
    gs.name = "actor"
    gid = gc.AddCollectors(B, 0, 0, gs)
    ins = register(gc, gid, "input", ["s", "legal", "id", "seq", "game_counter", "last_terminal", "move_idx"], True)
    outs = register(gc, gid, "reply", ["pi", "V"], True)
    gc.Start()
    last_seq = [-1] * N
    last_gc = [0] * N
    terminals = 0
    for it in range(150):
        infos = gc.Wait(0)
        s, legal = ins["s"][0], ins["legal"][0]
        # a "policy": prefers empty points (plane 9), mass also on illegal points -- the sampler must mask
        outs["pi"][0].copy_(s[:, 9].reshape(B, 361) + 0.01)
        h = {k: v.cpu().numpy() for k, v in ins.items()}
        for r in range(B):
            g = int(h["id"][0, r])
            if h["last_terminal"][0, r]:
                terminals += 1
                assert h["seq"][0, r] == 0 and h["game_counter"][0, r] == last_gc[g] + 1 and h["move_idx"][0, r] == 1
            else:
                assert h["seq"][0, r] == last_seq[g] + 1 and h["game_counter"][0, r] == last_gc[g]
            last_seq[g], last_gc[g] = int(h["seq"][0, r]), int(h["game_counter"][0, r])
            assert h["move_idx"][0, r] <= 61
        gc.Steps(infos)
    stats = gc.Stats()
    assert terminals >= N and stats[0] >= terminals and stats[8] > 0     # games counted, ply-cap hits
    gc.Stop()


def test_gcwrapper_offline_training_batches(go):
    """The Python caller's view (GCWrapper/Batch, train_df shape): device-resident batch tensors,
    callback receives batch.hist(0) views."""
    from elf_amd import go_loader
    args = go_loader.add_arguments().parse_args(
        ["--mode", "offline", "--num_games", "8", "--batchsize", "8", "--gpu", "0", "--seed", "7",
         "--list_file", os.path.join(GOLDEN, "corpus", "train.lst"), "--num_games_per_thread", "2",
         "--additional_labels", "id,seq,move_idx,aug_code,game_record_idx,winner"])
    GC = go_loader.initialize(args)
    seen = []

    def train(batch):
        b = batch.hist(0)
        assert b["s"].is_cuda and tuple(b["s"].shape) == (8, 25, 19, 19) and tuple(b["offline_a"].shape) == (8, 3)
        # stones planes are consistent with the empty plane
        assert torch.all(b["s"][:, 7] + b["s"][:, 8] + b["s"][:, 9] == 1)
        assert int(b["offline_a"].min()) >= 0 and int(b["offline_a"].max()) < 361
        seen.append(b["move_idx"].cpu().tolist())

    GC.reg_callback("train", train)
    GC.Start()
    for _ in range(20):
        GC.Run()
    GC.Stop()
    assert len(seen) == 20 and GC.params["num_action"] == 361


def _selfplay_context(go, N, B, ncoll, seed, fuse, keys, ply_cap=40):
    co = go.ContextOptions()
    co.num_games = N
    opt = go.GameOptions()
    opt.mode, opt.seed, opt.signal_restart, opt.end_on_two_passes, opt.ply_cap = "selfplay", seed, True, True, ply_cap
    opt.fuse_selfplay = fuse
    gc = go.GameContext(co, opt)
    gs = gc.CreateGroupStat()
    gs.name = "actor"
    gids = [gc.AddCollectors(B, 0, 0, gs) for _ in range(ncoll)]
    ins = [register(gc, gid, "input", keys, True) for gid in gids]
    outs = [register(gc, gid, "reply", ["pi", "V"], True) for gid in gids]
    gc.Start()
    return gc, gids, ins, outs


@pytest.mark.parametrize("N,B,ncoll,fuse_max", [(16, 16, 1, None), (16, 8, 2, None), (24, 8, 2, None), (10, 16, 1, None),
                                                 (16, 16, 1, "0")])
def test_fused_selfplay_step_against_oracle(go, monkeypatch, N, B, ncoll, fuse_max):
    """Selfplay with device tensors and a pi-only reply: Steps() is ONE launch (goenv_ply_policy_sel) that
    samples the move, applies it and writes the games' next samples into the collector's tensors.  Every
    sampled move is predicted by the oracle's sampler from the same pi, every delivered row (planes, mask,
    sequencing record) is checked against an oracle board, and the whole stream equals the unfused path's
    (Wait extracts, Steps samples + plies: three launches).  fuse_max "0" forces the form large selections
    take: the sampler as a launch of its own in front of the ply kernel."""
    if fuse_max is not None:
        monkeypatch.setenv("GOENV_FUSE_MAX", fuse_max)       # read by goenv_create
    seed, cap = 13, 40
    iters = 50 * -(-N // B)                                        # every game passes the ply cap at least once
    keys = ["s", "legal"] + LABELS
    streams = []
    for fuse in (True, False):
        gc, gids, ins, outs = _selfplay_context(go, N, B, ncoll, seed, fuse, keys, cap)
        oracles = [OracleBoard() for _ in range(N)]
        game = [0] * N
        seq = [0] * N
        fresh = [False] * N
        served = [0] * N
        rng = np.random.RandomState(5)
        rows = []
        for it in range(iters):
            infos = gc.Wait(0)
            k = gids.index(infos.gid)
            n = len(infos.s)
            assert n == min(B, N)
            h = {key: v.cpu().numpy()[0] for key, v in ins[k].items()}
            pi = rng.rand(n, 361).astype(np.float32)
            pi[0] = 0.0                                           # all-zero row: the cfg-2 fallback policy
            outs[k]["pi"][0, :n].copy_(torch.from_numpy(pi))
            for r, g in enumerate(infos.s):
                o = oracles[g]
                assert h["id"][r] == g and h["seq"][r] == seq[g] and h["game_counter"][r] == game[g]
                assert h["last_terminal"][r] == (1 if fresh[g] else 0)
                assert h["move_idx"][r] == o.canonical()["scalars"][0]
                assert np.array_equal(h["legal"][r], o.legal())
                assert_planes_equal(h["s"][r], o.extract(0), "iter %d game %d" % (it, g))
                rows.append((it, g, int(h["seq"][r]), fnv_planes(FNV_SEED, h["s"][r].reshape(-1))))
                seq[g] += 1
                fresh[g] = False
                served[g] += 1
                a = o.sample_policy(seed, g, game[g], pi[r], 0)
                ok, end = o.forward(a)
                assert ok
                if end or o.canonical()["scalars"][0] > cap:      # scored, counted, restarted in place
                    oracles[g] = OracleBoard()
                    game[g] += 1
                    seq[g] = 0
                    fresh[g] = True
            gc.Steps(infos)
        assert max(served) - min(served) <= 1, "games are not served evenly: %s" % served
        assert sum(game) > 0
        assert gc.Stats()[0] >= sum(game)
        gc.Stop()
        streams.append(rows)
    assert streams[0] == streams[1]


@pytest.mark.parametrize("device", [False, True])
def test_history_windows_equal_the_reference_module(go, device):
    """History length T > 1 (elf/hist.h:19-66, gate elf/comm_template.h:68-91): every delivered row is the
    window of the game's last T samples, oldest first, consecutive windows overlapping by one -- against
    tests/golden/batch_hist_streams.json, recorded from the compiled reference module with ContextOptions.T =
    GroupStat.hist_len = T and [T, B, ...] tensors."""
    gold = load_golden("batch_hist_streams.json")
    for run in gold["runs"]:
        cfg = run["cfg"]
        T, B = cfg["T"], cfg["num_games"]
        co = go.ContextOptions()
        co.num_games, co.T = B, T
        opt = go.GameOptions()
        opt.seed, opt.mode = cfg["seed"], "offline"
        opt.list_filename = os.path.join(GOLDEN, "corpus", cfg["lst"])
        opt.data_aug, opt.start_ratio_pre_moves, opt.ratio_pre_moves = cfg["data_aug"], cfg["start_ratio"], cfg["ratio"]
        opt.move_cutoff, opt.num_games_per_thread, opt.num_future_actions = cfg["cutoff"], cfg["games_per_thread"], cfg["k"]
        gc = go.GameContext(co, opt)
        gs = gc.CreateGroupStat()
        gs.hist_len = T
        gid = gc.AddCollectors(B, 0, 0, gs)
        bufs = {}
        for key in ["s", "offline_a"] + LABELS:
            info = gc.GetTensorSpec(gid, key, T)
            assert info.sz[:2] == [T, B]
            if device:
                v = torch.zeros(*info.sz, dtype=TORCH_TYPES[info.type], device="cuda:0")
                info.p, info.byte_size = v.data_ptr(), v.numel() * v.element_size()
            else:
                v = np.zeros(info.sz, dtype=NP_TYPES[info.type])
                info.p, info.byte_size = v.ctypes.data, v.nbytes
            gc.AddTensor(gid, "input", info)
            bufs[key] = v
        gc.Start()
        n_check = min(cfg["samples"], 25 if device else cfg["samples"])
        streams = {g: [] for g in range(B)}
        while min(len(v) for v in streams.values()) < n_check:
            infos = gc.Wait(0)
            assert infos.gid == gid
            h = {k: host(v) for k, v in bufs.items()}
            for r in range(len(infos.s)):
                g = int(h["id"][T - 1, r])
                assert g == infos.s[r]
                streams[g].append([[int(h[k][t, r]) for k in LABELS] + [[int(a) for a in h["offline_a"][t, r]]] +
                                   ["%016x" % fnv_planes(FNV_SEED, h["s"][t, r].reshape(-1))] for t in range(T)])
            gc.Steps(infos)
        gc.Stop()
        for g in range(B):
            want, got = run["streams"][str(g)][:n_check], streams[g][:n_check]
            for i, (a, b) in enumerate(zip(got, want)):
                assert a == b, "run %s game %d window %d: got %s, reference %s" % (cfg["name"], g, i, a, b)


def test_selfplay_sample_goes_to_an_actor_and_a_history_group(go):
    """The RL-training topology of ELF (docs/source/wrapper-python.rst:38-65): every sample goes to the `actor`
    group (T = 1, replies a) AND to an unnamed `train` group with history T = 4 (keys s, a, labels).  The
    game moves only when both are done with it.  Train windows: T consecutive samples of one game, planes of
    entry t = the oracle's at that time, a[t] = the action replied for that sample, windows overlap by one."""
    N, B, T = 8, 8, 4
    co = go.ContextOptions()
    co.num_games, co.T = N, T
    opt = go.GameOptions()
    opt.mode, opt.seed = "selfplay", 3
    gc = go.GameContext(co, opt)
    ga = gc.CreateGroupStat()
    ga.name = "actor"
    gid_a = gc.AddCollectors(B, 0, 0, ga)
    gt = gc.CreateGroupStat()
    gt.hist_len = T
    gid_t = gc.AddCollectors(B, 1, 0, gt)

    def reg(gid, which, keys, T_):
        out = {}
        for key in keys:
            info = gc.GetTensorSpec(gid, key, T_)
            v = torch.zeros(*info.sz, dtype=TORCH_TYPES[info.type], device="cuda:0")
            info.p, info.byte_size = v.data_ptr(), v.numel() * v.element_size()
            gc.AddTensor(gid, which, info)
            out[key] = v
        return out

    act_in = reg(gid_a, "input", ["s", "legal", "id", "seq"], 1)
    act_out = reg(gid_a, "reply", ["a", "V"], 1)
    tr_in = reg(gid_t, "input", ["s", "a", "id", "seq", "move_idx"], T)
    gc.Start()
    oracles = [OracleBoard() for _ in range(N)]
    planes = {}                      # (game, seq) -> planes the oracle had when that sample was taken
    action = {}                      # (game, seq) -> action replied
    rng = np.random.RandomState(1)
    windows = {g: [] for g in range(N)}
    moved = [0] * N                  # moves applied = seq of the game's current sample
    n_actor = n_train = 0
    for it in range(60):
        infos = gc.Wait(0)
        assert infos.gid in (gid_a, gid_t)
        if infos.gid == gid_a:
            n_actor += 1
            h = {k: v.cpu().numpy()[0] for k, v in act_in.items()}
            acts = []
            for r, g in enumerate(infos.s):
                o = oracles[g]
                assert h["id"][r] == g
                seq = int(h["seq"][r])
                assert_planes_equal(h["s"][r], o.extract(0), "actor game %d seq %d" % (g, seq))
                planes[(g, seq)] = o.extract(0).copy()
                legal = np.flatnonzero(o.legal())
                a = int(legal[rng.randint(legal.size)])
                action[(g, seq)] = a
                acts.append(a)
                assert o.forward(a)[0]
                moved[g] += 1
            act_out["a"].copy_(torch.tensor([acts], dtype=torch.int64))
        else:
            n_train += 1
            h = {k: v.cpu().numpy() for k, v in tr_in.items()}
            for r, g in enumerate(infos.s):
                seqs = [int(h["seq"][t, r]) for t in range(T)]
                assert seqs == list(range(seqs[0], seqs[0] + T)), seqs
                assert all(int(h["id"][t, r]) == g for t in range(T))
                for t in range(T):
                    if (g, seqs[t]) not in planes:       # the train group got the newest sample before the actor did
                        assert t == T - 1 and seqs[t] == moved[g]
                        planes[(g, seqs[t])] = oracles[g].extract(0).copy()
                    assert_planes_equal(h["s"][t, r], planes[(g, seqs[t])], "train game %d seq %d" % (g, seqs[t]))
                    assert int(h["move_idx"][t, r]) == seqs[t] + 1
                    if (g, seqs[t]) in action:   # the newest entry's action may not have been replied yet
                        assert int(h["a"][t, r]) == action[(g, seqs[t])]
                windows[g].append(seqs)
        gc.Steps(infos)
    gc.Stop()
    assert n_train >= 10 and n_actor >= 3 * n_train - 3
    for g in range(N):
        w = windows[g]
        assert w[0][0] == 0
        for a, b in zip(w, w[1:]):
            assert b[0] == a[-1], "windows of game %d do not overlap by one: %s %s" % (g, a, b)


def test_gcwrapper_offline_with_history(go):
    """train_df shape with --T 3 (the reference's ContextArgs default is T = 6, elf/context_utils.py:20):
    batch.hist(t) is entry t of every row's window; consecutive entries are consecutive samples."""
    from elf_amd import go_loader
    args = go_loader.add_arguments().parse_args(
        ["--mode", "offline", "--num_games", "4", "--batchsize", "4", "--T", "3", "--gpu", "0", "--seed", "7",
         "--list_file", os.path.join(GOLDEN, "corpus", "train.lst"), "--num_games_per_thread", "1",
         "--additional_labels", "id,seq,move_idx,game_record_idx"])
    GC = go_loader.initialize(args)
    seen = []

    def train(batch):
        assert tuple(batch["s"].shape) == (3, 4, 25, 19, 19)
        seq = torch.stack([batch.hist(t)["seq"] for t in range(3)]).cpu()
        assert torch.equal(seq[1], seq[0] + 1) and torch.equal(seq[2], seq[1] + 1)
        rec = torch.stack([batch.hist(t)["game_record_idx"] for t in range(3)]).cpu()
        mi = torch.stack([batch.hist(t)["move_idx"] for t in range(3)]).cpu()
        same = rec[1] == rec[0]                      # same record: the next sample is one ply later
        assert torch.equal(mi[1][same], mi[0][same] + 1)
        seen.append(seq[2].tolist())

    GC.reg_callback("train", train)
    GC.Start()
    for _ in range(12):
        GC.Run()
    GC.Stop()
    assert len(seen) == 12 and seen[1][0] == seen[0][0] + 2      # windows advance by T - 1


def test_offline_sample_taken_by_two_groups(go):
    """Two unnamed exclusive groups (T = 1): every sample goes to BOTH, the game moves on when both have been
    stepped (elf/comm_template.h:229-257).  Each group's per-game stream must be the golden stream of the
    reference module, whatever the order in which the two groups' batches are served."""
    run = load_golden("batch_streams.json")["runs"][0]
    cfg = run["cfg"]
    B = cfg["num_games"]
    co = go.ContextOptions()
    co.num_games = B
    opt = go.GameOptions()
    opt.seed, opt.mode = cfg["seed"], "offline"
    opt.list_filename = os.path.join(GOLDEN, "corpus", cfg["lst"])
    opt.data_aug, opt.start_ratio_pre_moves, opt.ratio_pre_moves = cfg["data_aug"], cfg["start_ratio"], cfg["ratio"]
    opt.move_cutoff, opt.num_games_per_thread, opt.num_future_actions = cfg["cutoff"], cfg["games_per_thread"], cfg["k"]
    gc = go.GameContext(co, opt)
    gids, bufs = [], []
    for e in range(2):
        gid = gc.AddCollectors(B, e, 0, gc.CreateGroupStat())
        gids.append(gid)
        bufs.append(register(gc, gid, "input", ["s", "offline_a"] + LABELS, e == 0))   # one on the device, one on the host
    gc.Start()
    n_check = 60
    streams = [{g: [] for g in range(B)} for _ in range(2)]
    served = [0, 0]
    while min(len(v) for st in streams for v in st.values()) < n_check:
        infos = gc.Wait(0)
        k = gids.index(infos.gid)
        served[k] += 1
        assert abs(served[0] - served[1]) <= 1                      # a game waits for both takers
        h = {key: host(v) for key, v in bufs[k].items()}
        for r in range(len(infos.s)):
            g = int(h["id"][0, r])
            streams[k][g].append([int(h[key][0, r]) for key in LABELS] + [[int(a) for a in h["offline_a"][0, r]]] +
                                 ["%016x" % fnv_planes(FNV_SEED, h["s"][0, r].reshape(-1))])
        gc.Steps(infos)
    gc.Stop()
    for k in range(2):
        for g in range(B):
            assert streams[k][g][:n_check] == run["streams"][str(g)][:n_check], (k, g)
