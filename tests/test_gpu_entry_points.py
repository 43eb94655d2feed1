"""GPU parity tests of the remaining C-ABI entry points (through elf_amd/goenv.py): policy
sampling, move-list replay, selections, sequencing records.  Bit-exact against the CPU oracle."""
import numpy as np
import pytest

from common import assert_canon_equal, assert_planes_equal, load_golden
from oracle.pyoracle import OracleBoard

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("waves_per_board")]
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def goenv():
    from elf_amd import goenv as m
    m.load_library()
    assert torch.cuda.is_available()
    return m


def midgame(goenv, B, seed, plies):
    """B boards advanced by the on-device policy, with their oracle twins."""
    env = goenv.GoEnv(B, seed=seed, max_ply=722)
    acts = env.new(torch.int32)
    oracles = [OracleBoard() for _ in range(B)]
    for _ in range(plies):
        env.ply(goenv.F_RANDOM_POLICY, actions_out=acts)
        for b, a in enumerate(acts.cpu().tolist()):
            assert oracles[b].forward(a)[0]
    return env, oracles


def test_sample_policy_matches_oracle(goenv):
    B, seed = 40, 77
    env, oracles = midgame(goenv, B, seed, 90)
    rng = np.random.RandomState(3)
    pi = rng.rand(B, 361).astype(np.float32)
    pi[1] = 0.0                                  # all-zero row: falls back to the cfg-2 policy
    pi[2] = -1.0                                 # nothing positive: same
    pi[3] *= 5.0                                 # weights above 1 saturate at 2^24
    pi[4, :] = 0.0
    pi[4, np.flatnonzero(oracles[4].legal())[:3]] = [1e-9, 0.5, 2e-8]   # tiny weights floor to 0
    pi[5] = np.where(oracles[5].legal() > 0, 0.0, 1.0)                  # all mass on illegal points
    augs = (np.arange(B) % 8).astype(np.int32)
    out = torch.zeros(B, dtype=torch.int64, device=env.device)
    env.sample_policy(torch.from_numpy(pi).to(env.device), out, aug_codes=torch.from_numpy(augs).to(env.device))
    got = out.cpu().tolist()
    for b, o in enumerate(oracles):
        want = o.sample_policy(seed, b, 0, pi[b], int(augs[b]))
        assert got[b] == want, "board %d: GPU %d, oracle %d" % (b, got[b], want)
        assert got[b] == -1 or o.legal()[got[b]] == 1
    # a selection: rows follow idx
    idx = torch.tensor([7, 3, 30], dtype=torch.int32, device=env.device)
    out3 = torch.zeros(3, dtype=torch.int64, device=env.device)
    env.sample_policy(torch.from_numpy(pi[[7, 3, 30]]).to(env.device), out3, idx=idx,
                      aug_codes=torch.from_numpy(augs[[7, 3, 30]]).to(env.device))
    assert out3.cpu().tolist() == [got[7], got[3], got[30]]
    env.close()


def test_replay_move_lists_and_stop_at_illegal(goenv):
    g = load_golden("playouts.json")
    games = g["games"][:6]
    B = len(games) + 2
    env = goenv.GoEnv(B, seed=1)
    stride = 700
    moves = np.full((B, stride), -1, np.int16)
    nm = np.zeros(B, np.int32)
    want_applied = []
    oracles = [OracleBoard() for _ in range(B)]
    for b, game in enumerate(games):
        mv = list(game["moves"])
        if b % 2 == 1:                            # corrupt: repeat an earlier (occupied) point
            cut = 40 + 10 * b
            mv = mv[:cut] + [mv[5]] + mv[cut:]
        mv = mv[:stride]
        moves[b, :len(mv)] = mv
        nm[b] = len(mv)
        n = 0
        for a in mv:
            if not oracles[b].forward(a)[0]:
                break
            n += 1
        want_applied.append(n)
    nm[B - 2] = 0                                 # empty list: board untouched
    want_applied.append(0)
    moves[B - 1, :3] = [-1, 60, -1]               # passes are moves too
    nm[B - 1] = 3
    for a in (-1, 60, -1):
        oracles[B - 1].forward(a)
    want_applied.append(3)
    napplied = env.new(torch.int32)
    env.replay(torch.from_numpy(moves).to(env.device), torch.from_numpy(nm).to(env.device), napplied)
    assert napplied.cpu().tolist() == want_applied
    assert any(w < n for w, n in zip(want_applied, nm.tolist()))
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    env.features(feat, None, legal)
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    for b, o in enumerate(oracles):
        assert_canon_equal({k: v[b] for k, v in st.items()}, o.canonical(), "board %d" % b)
        assert np.array_equal(legal[b].cpu().numpy(), o.legal())
        assert_planes_equal(feat[b].cpu().numpy(), o.extract(0), "board %d" % b)
    env.close()


def test_selections_and_sequencing_records(goenv):
    """ply_sel / collect_sel on an index list: rows follow the list, untouched boards stay
    untouched; sequencing record in both modes (episodes / the reference's run-on selfplay)."""
    B, seed = 16, 5
    env, oracles = midgame(goenv, B, seed, 30)
    dev = env.device
    sel = [11, 2, 7, 13]
    idx = torch.tensor(sel, dtype=torch.int32, device=dev)
    n = len(sel)
    meta = dict(id=torch.zeros(n, dtype=torch.int32, device=dev), seq=torch.zeros(n, dtype=torch.int32, device=dev),
                game_counter=torch.zeros(n, dtype=torch.int32, device=dev),
                last_terminal=torch.zeros(n, dtype=torch.int8, device=dev),
                move_idx=torch.zeros(n, dtype=torch.int32, device=dev),
                winner=torch.ones(n, dtype=torch.int32, device=dev),
                aug_code=torch.zeros(n, dtype=torch.int32, device=dev),
                game_record_idx=torch.zeros(n, dtype=torch.int32, device=dev))
    feat = torch.zeros(n, 25, 19, 19, device=dev)
    legal = torch.zeros(n, 361, dtype=torch.uint8, device=dev)
    augs = torch.tensor([0, 5, 2, 7], dtype=torch.int32, device=dev)
    for rep in range(3):
        env.collect_sel(idx=idx, aug_codes=augs, feat_out=feat, legal_out=legal, meta=meta)
        assert meta["id"].cpu().tolist() == sel and meta["seq"].cpu().tolist() == [rep] * n
        assert meta["game_counter"].cpu().tolist() == [0] * n and meta["last_terminal"].cpu().tolist() == [0] * n
        assert meta["move_idx"].cpu().tolist() == [31] * n and meta["winner"].cpu().tolist() == [0] * n
        assert meta["aug_code"].cpu().tolist() == [0, 5, 2, 7] and meta["game_record_idx"].cpu().tolist() == [-1] * n
    for r, b in enumerate(sel):
        assert_planes_equal(feat[r].cpu().numpy(), oracles[b].extract(int(augs[r])), "row %d" % r)
        assert np.array_equal(legal[r].cpu().numpy(), oracles[b].legal())
    # an illegal action with RESET_ILLEGAL restarts that game only; the others play on
    acts = []
    for r, b in enumerate(sel):
        lg = np.flatnonzero(oracles[b].legal())
        acts.append(int(np.flatnonzero(oracles[b].canonical()["color"])[0]) if r == 1 else int(lg[0]))
    status = torch.zeros(n, dtype=torch.int32, device=dev)
    env.ply_sel(idx=idx, flags=goenv.F_RESET_ILLEGAL, actions=torch.tensor(acts, dtype=torch.int64, device=dev),
                status_out=status)
    assert status.cpu().tolist() == [1, 0, 1, 1]
    for r, b in enumerate(sel):
        if r == 1:
            oracles[b] = OracleBoard()
        else:
            assert oracles[b].forward(acts[r])[0]
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    for b, o in enumerate(oracles):
        assert_canon_equal({k: v[b] for k, v in st.items()}, o.canonical(), "board %d" % b)
    # episodes mode: the restarted game reports a new episode; run-on mode keeps counting
    env.collect_sel(idx=idx, meta=meta)
    assert meta["seq"].cpu().tolist() == [3, 0, 3, 3] and meta["game_counter"].cpu().tolist() == [0, 1, 0, 0]
    assert meta["last_terminal"].cpu().tolist() == [0, 1, 0, 0] and meta["move_idx"].cpu().tolist() == [32, 1, 32, 32]
    so = goenv.SampleOut.from_tensors(meta, mode=1)
    import ctypes as C
    env._chk(env.lib.goenv_collect_sel(env._h, C.c_void_p(idx.data_ptr()), 0, n, None, None, None, C.byref(so),
                                       env._stream()))
    assert meta["seq"].cpu().tolist() == [4, 4, 4, 4] and meta["game_counter"].cpu().tolist() == [0] * n
    assert meta["last_terminal"].cpu().tolist() == [0] * n
    sub = {k: v.cpu().numpy() for k, v in env.get_state_sel(idx=idx).items()}
    for r, b in enumerate(sel):
        assert_canon_equal({k: v[r] for k, v in sub.items()}, oracles[b].canonical(), "sel row %d" % r)
    # errors: out-of-range selection is reported, not fatal
    with pytest.raises(goenv.GoEnvError):
        env.collect_sel(first=10, count=10, feat_out=feat)
    env.close()


@pytest.mark.parametrize("gen", ["default", "two_waves", "pipelined_unpaced", "pipelined_pinned_pace"])
def test_generate_equals_per_ply_launches(goenv, monkeypatch, gen):
    """The persistent generator writes, slot by slot, exactly what the per-ply launches write
    (moves, masks, planes), restarts included, and ends in the same state and statistics -- through
    each of its kernels: k_generate_pipe with the self-tuning rate limiter (the default up to the
    number of resident boards), k_generate<2> (GOENV_GEN_PIPE=0), the pipelined kernel without and
    with a pinned pace."""
    env_vars = {"default": {}, "two_waves": {"GOENV_GEN_PIPE": "0"},
                "pipelined_unpaced": {"GOENV_GEN_PIPE": "1", "GOENV_GEN_PACE_NS": "0"},
                "pipelined_pinned_pace": {"GOENV_GEN_PIPE": "1", "GOENV_GEN_PACE_NS": "9000"}}[gen]
    for k, v in env_vars.items():
        monkeypatch.setenv(k, v)     # read by goenv_create
    B, T, seed = 48, 40, 21
    rounds = 4                       # a low ply cap forces restarts inside the rings
    a = goenv.GoEnv(B, seed=seed, max_ply=60)
    b = goenv.GoEnv(B, seed=seed, max_ply=60)
    dev = a.device
    augs = (torch.arange(B, dtype=torch.int32, device=dev) % 8).contiguous()
    feat_ring = torch.zeros(T, B, 25, 19, 19, device=dev)
    legal_ring = torch.zeros(T, B, 361, dtype=torch.uint8, device=dev)
    act_ring = torch.zeros(T, B, dtype=torch.int32, device=dev)
    feat, legal, acts = b.new(torch.float32, 25, 19, 19), b.new(torch.uint8, 361), b.new(torch.int32)
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    for r in range(rounds):
        a.generate(T, feat_ring, legal_ring, act_ring, aug_codes=augs)
        for t in range(T):
            b.ply(flags, aug_codes=augs, feat_out=feat, legal_out=legal, actions_out=acts)
            assert torch.equal(act_ring[t], acts), (r, t)
            assert torch.equal(legal_ring[t], legal), (r, t)
            assert torch.equal(feat_ring[t].view(torch.int32), feat.view(torch.int32)), (r, t)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        x, y = (sa[k].view(torch.int16), sb[k].view(torch.int16)) if sa[k].dtype == torch.uint16 else (sa[k], sb[k])
        assert torch.equal(x, y), k
    assert a.stats() == b.stats() and a.stats()[goenv.ST_GAMES] > B
    a.close()
    b.close()


@pytest.mark.parametrize("B", [1, 7, 9, 13, 100])
def test_row_dealing_per_xcd_is_invisible(goenv, monkeypatch, B):
    """Which workgroup takes which output row (xcd_row: every XCD a contiguous run of rows) must not show in
    any output: ply, feature and generator kernels with the map on equal the same calls with it off, for
    board counts that are not multiples of 8 (one board, fewer boards than XCDs, ragged last runs)."""
    T, seed = 5, 31
    envs = []
    for m in ("0", "1"):
        monkeypatch.setenv("GOENV_XCD_MAP", m)          # read by goenv_create
        envs.append(goenv.GoEnv(B, seed=seed, max_ply=40))
    monkeypatch.delenv("GOENV_XCD_MAP")
    dev = envs[0].device
    augs = ((torch.arange(B, dtype=torch.int32, device=dev) * 5) % 8).contiguous()
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    outs = []
    for env in envs:
        got = []
        ring = (torch.zeros(T, B, 25, 19, 19, device=dev), torch.zeros(T, B, 361, dtype=torch.uint8, device=dev),
                torch.zeros(T, B, dtype=torch.int32, device=dev))
        feat, legal, acts = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361), env.new(torch.int32)
        for _ in range(12):
            env.generate(T, *ring, aug_codes=augs)
            got += [t.clone() for t in ring]
            env.ply(flags, aug_codes=augs, feat_out=feat, legal_out=legal, actions_out=acts)
            got += [feat.clone(), legal.clone(), acts.clone()]
            env.features(feat, augs, legal)
            got += [feat.clone(), legal.clone()]
        got += [v for _, v in sorted(env.get_state().items())]
        outs.append(got)
        env.close()
    for i, (x, y) in enumerate(zip(*outs)):
        if x.dtype == torch.float32:
            x, y = x.view(torch.int32), y.view(torch.int32)
        elif x.dtype == torch.uint16:
            x, y = x.view(torch.int16), y.view(torch.int16)
        assert torch.equal(x, y), i


def test_copy_boards_branches_positions(goenv):
    """Tree-search style branching: children = copies of a parent, each stepped with its own move."""
    B, seed = 12, 8
    env, oracles = midgame(goenv, B, seed, 50)
    dev = env.device
    parent = 3
    kids = [6, 7, 8, 9]
    env.copy_boards(torch.tensor([parent] * 4, dtype=torch.int32, device=dev),
                    torch.tensor(kids, dtype=torch.int32, device=dev))
    moves = [int(m) for m in np.flatnonzero(oracles[parent].legal())[:3]] + [-1]
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    feat = torch.zeros(4, 25, 19, 19, device=dev)
    env.ply_sel(idx=torch.tensor(kids, dtype=torch.int32, device=dev),
                actions=torch.tensor(moves, dtype=torch.int64, device=dev), feat_out=feat, status_out=status)
    assert status.cpu().tolist() == [1, 1, 1, 1]
    for k, mv in zip(kids, moves):
        oracles[k] = oracles[parent].clone()
        assert oracles[k].forward(mv)[0]
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    for b, o in enumerate(oracles):                      # parent and bystanders untouched
        assert_canon_equal({k: v[b] for k, v in st.items()}, o.canonical(), "board %d" % b)
    for r, k in enumerate(kids):
        assert_planes_equal(feat[r].cpu().numpy(), oracles[k].extract(0), "child %d" % k)
    with pytest.raises(goenv.GoEnvError):
        env.copy_boards(torch.zeros(B + 1, dtype=torch.int32, device=dev), torch.zeros(B + 1, dtype=torch.int32, device=dev))
    env.close()


def test_restart_then_rejected_action_counts_the_game_once(goenv):
    """A finished board under AUTO_RESTART is scored, counted and restarted BEFORE the external action is
    validated; if the action is then rejected the restarted record must still be stored, or the next call
    would count the same game again."""
    env = goenv.GoEnv(4, seed=2, max_ply=722)
    dev = env.device
    two_passes = torch.full((4,), -1, dtype=torch.int64, device=dev)
    env.ply(0, actions=two_passes)
    env.ply(0, actions=two_passes)                      # every board: IsGameEnd
    assert env.get_state()["scalars"][:, goenv.S_DONE].cpu().tolist() == [1, 1, 1, 1]
    bad = torch.full((4,), 361, dtype=torch.int64, device=dev)      # out of range: rejected
    status = env.new(torch.int32)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    for _ in range(2):
        env.ply(goenv.F_AUTO_RESTART, actions=bad, feat_out=feat, legal_out=legal, status_out=status)
        assert status.cpu().tolist() == [0, 0, 0, 0]
    st = env.stats()
    assert st[goenv.ST_GAMES] == 4 and st[goenv.ST_PLIES] == 8
    sc = env.get_state()["scalars"].cpu()
    assert sc[:, goenv.S_DONE].tolist() == [0] * 4 and sc[:, goenv.S_PLY].tolist() == [1] * 4
    assert sc[:, goenv.S_GAME].tolist() == [1] * 4
    o = OracleBoard()                                    # what was written belongs to the restarted (empty) board
    for b in range(4):
        assert_planes_equal(feat[b].cpu().numpy(), o.extract(0), "board %d" % b)
        assert np.array_equal(legal[b].cpu().numpy(), o.legal())
    env.close()


def test_copy_rows_gathers_and_scatters(goenv):
    env = goenv.GoEnv(2, seed=1)
    dev = env.device
    for shape, dtype in (((40, 361), torch.uint8), ((40, 25, 19, 19), torch.float32), ((40,), torch.int64)):
        src = (torch.rand(shape, device=dev) * 200).to(dtype)
        dst = torch.zeros((25,) + shape[1:], dtype=dtype, device=dev)
        sr = torch.tensor([3, 39, 0, 17, 3], dtype=torch.int32, device=dev)
        dr = torch.tensor([24, 1, 7, 0, 9], dtype=torch.int32, device=dev)
        env.copy_rows(src, dst, sr, dr)
        want = torch.zeros_like(dst)
        want[dr.long()] = src[sr.long()]
        assert torch.equal(dst, want)
        env.copy_rows(src, dst, None, None, n=25)        # identity lists
        assert torch.equal(dst, src[:25])
    with pytest.raises(goenv.GoEnvError):
        env.copy_rows(torch.zeros(4, 8, device=dev), torch.zeros(4, 9, device=dev))
    env.close()


def _pack_offline_rows(buf, rows, k):
    """rows: dicts(board, aug, reset, handi, moves, labels[8], offline_a[k]) -> the block layout of
    goenv_offline_batch (include/goenv.h); returns moves_offset."""
    n, row_w = len(rows), 12 + 2 * k
    words = buf[:n * row_w * 4].view(np.int32).reshape(n, row_w)
    moves_off = n * row_w * 4
    mv = buf[moves_off:].view(np.int16)
    at = 0
    for r, d in enumerate(rows):
        words[r, 0] = d["board"]
        words[r, 1] = d["aug"] | (8 if d["reset"] else 0) | (d["handi"] << 4) | (len(d["moves"]) << 8)
        words[r, 2] = at
        words[r, 3] = 0
        words[r, 4:12] = d["labels"]
        words[r, 12:].view(np.int64)[:] = d["offline_a"]
        mv[at:at + len(d["moves"])] = d["moves"]
        at += len(d["moves"])
    return moves_off


@pytest.mark.parametrize("device_block", [False, True])
def test_offline_batch_through_the_c_abi(goenv, device_block):
    """goenv_offline_batch: per row an optional reset + handicap, the row's record moves, then planes under the
    row's D4 code, mask and the host's labels -- one launch -- against oracle boards doing the same
    (OfflineLoader reload / next / extract, go/offpolicy_loader.cc:70-145).  Rows are read in place from a
    block of the pinned staging ring, or from device memory."""
    g = load_golden("sample1.json")
    rec = g["moves"][:120]
    B, k, seed = 37, 3, 5
    env, oracles = midgame(goenv, B, seed, 40)
    rng = np.random.RandomState(9)
    perm = rng.permutation(B)
    bad_rows = 0
    for rnd in range(3):
        rows = []
        for r, b in enumerate(perm):
            b = int(b)
            kind = (r + rnd) % 4
            d = dict(board=b, aug=int(rng.randint(8)), reset=kind in (0, 1), handi=[0, 4, 0, 0][kind],
                     labels=[int(x) for x in rng.randint(-5, 1000, 8)], offline_a=[int(x) for x in rng.randint(0, 361, k)])
            d["labels"][3] = int(rng.randint(2))
            o = oracles[b]
            if d["reset"]:
                o.clear()
                o.apply_handicap(d["handi"])
                # after a handicap WHITE is to move: the record's first move then plays as white, as in the reference
                d["moves"] = rec[:int(rng.randint(0, 90))]
            elif kind == 2:
                d["moves"] = []                                       # nothing owed: planes of the position as it stands
            else:
                idx = np.flatnonzero(o.candidates())
                d["moves"] = [int(idx[rng.randint(len(idx))])] if len(idx) else [-1]
            applied = 0
            for m in d["moves"]:
                if not o.forward(int(m))[0]:
                    break
                applied += 1
            d["want_applied_all"] = applied == len(d["moves"])
            rows.append(d)
        nbytes = B * (12 + 2 * k) * 4 + 2 * sum(len(d["moves"]) for d in rows) + 64
        buf, ptr = env.stage_acquire(nbytes)
        moves_off = _pack_offline_rows(buf, rows, k)
        if device_block:
            dev = torch.from_numpy(buf.copy()).to(env.device)
            ptr_used = dev.data_ptr()
        else:
            ptr_used = ptr
        feat = torch.full((B, 25, 19, 19), -2.0, device=env.device)
        legal = torch.full((B, 361), 7, dtype=torch.uint8, device=env.device)
        labels = {kk: torch.zeros(B, dtype=goenv.SampleOut.DTYPES[kk], device=env.device) for kk in goenv.SampleOut.DTYPES}
        offa = torch.zeros(B, k, dtype=torch.int64, device=env.device)
        env.offline_batch(ptr_used, B, k, moves_off, feat, legal, labels, offa)
        if device_block:                                               # the ring block was not consumed: hand it back
            env.offline_batch(ptr, 0, k, moves_off)
        torch.cuda.synchronize()
        f, lg = feat.cpu().numpy().reshape(B, -1), legal.cpu().numpy()
        lab = {kk: v.cpu().numpy() for kk, v in labels.items()}
        oa = offa.cpu().numpy()
        st = {kk: v.cpu().numpy() for kk, v in env.get_state().items()}
        names = ["id", "seq", "game_counter", "last_terminal", "move_idx", "winner", "aug_code", "game_record_idx"]
        for r, d in enumerate(rows):
            o = oracles[d["board"]]
            w = "round %d row %d board %d" % (rnd, r, d["board"])
            assert_planes_equal(f[r], o.extract(d["aug"]), w)
            assert np.array_equal(lg[r], o.legal()), w
            assert_canon_equal({kk: v[d["board"]] for kk, v in st.items()}, o.canonical(), w)
            assert [int(lab[nm][r]) for nm in names] == d["labels"], w
            assert oa[r].tolist() == d["offline_a"], w
        bad_rows += sum(0 if d["want_applied_all"] else 1 for d in rows)
        # (the sample1 prefix is legal from an empty board, not always after four handicap stones: such a row
        # stops at the rejected move, exactly as the oracle did above, and is counted)
        assert env.offline_bad_moves() == bad_rows
    # argument errors: a row naming a board outside the context, moves past the end of the block
    buf, ptr = env.stage_acquire(4096)
    rows = [dict(board=B + 3, aug=0, reset=False, handi=0, moves=[], labels=[0] * 8, offline_a=[0] * k)]
    off = _pack_offline_rows(buf, rows, k)
    with pytest.raises(goenv.GoEnvError, match="outside the context"):
        env.offline_batch(ptr, 1, k, off)
    rows[0]["board"] = 0
    off = _pack_offline_rows(buf, rows, k)
    buf[:8].view(np.int32)[1] = 1 << 30                                 # claims 4 M moves
    with pytest.raises(goenv.GoEnvError, match="past the end"):
        env.offline_batch(ptr, 1, k, off)
    env.close()


@pytest.mark.gpu
def test_rotating_outputs_same_planes_as_one_buffer(goenv):
    """Launches of up to 4,096 boards store their planes write-through while what the last launches wrote fits the
    Infinity Cache, and with plain stores once a caller rotates through more output than that (planes_in_flight in
    goenv.hip): 640 boards into one buffer against 640 boards into a ring of eight (185 MB) -- the same bytes."""
    B, R = 640, 8
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    a, b = goenv.GoEnv(B, seed=5, max_ply=722), goenv.GoEnv(B, seed=5, max_ply=722)
    one, one_legal = a.new(torch.float32, 25, 19, 19), a.new(torch.uint8, 361)
    ring = torch.zeros(R, B, 25, 19, 19, device=b.device)
    ring_legal = torch.zeros(R, B, 361, dtype=torch.uint8, device=b.device)
    for t in range(40):
        a.ply(flags, feat_out=one, legal_out=one_legal)
        b.ply(flags, feat_out=ring[t % R], legal_out=ring_legal[t % R])
        assert torch.equal(one.view(torch.int32), ring[t % R].view(torch.int32)), "ply %d" % t
        assert torch.equal(one_legal, ring_legal[t % R])
    sa, sb = a.get_state(), b.get_state()
    assert all(torch.equal(sa[k], sb[k]) for k in sa)
    a.close()
    b.close()
