"""The pipelined generator kernel (k_generate_pipe: a rules wave and a feature wave per board, a ring of
snapshots in LDS between them, rate-limited plane stores) against the two-waves-per-board kernel
(GOENV_GEN_PIPE=0) and the CPU oracle: launches shorter than the snapshot ring, outputs left out, D4 codes,
the constant planes preset, a launch inside a captured graph, and the self-tuning pace (which may change
WHEN a sample is written, never a byte of it).  Bar: bit-exact."""
import numpy as np
import pytest

from common import assert_planes_equal
from oracle.pyoracle import OracleBoard

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def goenv():
    from elf_amd import goenv as m
    m.load_library()
    assert torch.cuda.is_available()
    return m


def make(goenv, monkeypatch, pipe, n, seed, max_ply=80, pace=None):
    monkeypatch.setenv("GOENV_GEN_PIPE", "1" if pipe else "0")
    if pace is not None:
        monkeypatch.setenv("GOENV_GEN_PACE_NS", str(pace))
    else:
        monkeypatch.delenv("GOENV_GEN_PACE_NS", raising=False)
    env = goenv.GoEnv(n, seed=seed, max_ply=max_ply)
    return env


def rings(env, T, what, augs):
    dev, n = env.device, env.n
    feat = torch.full((T, n, 25, 19, 19), -2.0, device=dev) if "f" in what else None
    legal = torch.full((T, n, 361), 5, dtype=torch.uint8, device=dev) if "l" in what else None
    acts = torch.full((T, n), -7, dtype=torch.int32, device=dev) if "a" in what else None
    return feat, legal, acts


def same(x, y):
    if x is None:
        return y is None
    if x.dtype == torch.float32:
        x, y = x.view(torch.int32), y.view(torch.int32)
    elif x.dtype == torch.uint16:
        x, y = x.view(torch.int16), y.view(torch.int16)
    return torch.equal(x, y)


@pytest.mark.parametrize("n,T,what,aug", [(64, 1, "fla", False), (37, 2, "fla", True), (100, 3, "fl", True),
                                          (96, 5, "la", False), (50, 9, "a", False), (130, 11, "f", True),
                                          (300, 23, "fla", True),
                                          # more boards than resident workgroups: several boards per workgroup, waiting
                                          # in LDS (k_generate_pipe_multi), ragged last turn
                                          (2500, 5, "fla", True), (4100, 3, "fl", False), (8200, 2, "fla", True),
                                          # more boards than that kernel holds: launches over chunks of the boards, one
                                          # after the other (16,500 = 3 x 5,504: the last one shorter; 20,000: 3 x 6,672)
                                          (16500, 2, "fla", True), (20000, 1, "fl", False)])
def test_pipelined_equals_two_wave_kernel(goenv, monkeypatch, n, T, what, aug):
    out = []
    monkeypatch.setenv("GOENV_GEN_CHUNK_MIN_PLIES", "1")     # (by default launches of fewer than 8 plies are not cut into chunks)
    for pipe in (False, True):
        env = make(goenv, monkeypatch, pipe, n, seed=5, max_ply=80 if n < 2000 else 12)
        augs = ((torch.arange(n, dtype=torch.int32, device=env.device) * 3) % 8).contiguous() if aug else None
        got = []
        for _ in range(6):                              # launches follow one another: games end and restart inside
            r = rings(env, T, what, augs)
            env.generate(T, *r, aug_codes=augs)
            torch.cuda.synchronize()
            got += list(r)
        got += [v for _, v in sorted(env.get_state().items())]
        got.append(torch.tensor(env.stats()))
        assert env.generate_kernel() == ("k_generate" if not pipe else ("k_generate_pipe" if n <= 2048 else "k_generate_pipe_multi"))
        out.append(got)
        env.close()
        del r, got
        torch.cuda.empty_cache()
    for i, (x, y) in enumerate(zip(*out)):
        assert same(x, y), (i, n, T, what)
    assert int(out[0][-1][goenv.ST_GAMES]) > 0 or T * 6 < 80


def test_chunks_only_for_launches_of_eight_plies(goenv, monkeypatch):
    """More boards than k_generate_pipe_multi holds: launches of at least 8 plies go through it in chunks, shorter ones through
    k_generate (a launch boundary costs every chunk); the same rings either way."""
    n = 16500
    monkeypatch.delenv("GOENV_GEN_CHUNK_MIN_PLIES", raising=False)
    a, b = make(goenv, monkeypatch, True, n, seed=9, max_ply=20), make(goenv, monkeypatch, False, n, seed=9, max_ply=20)
    for T, kernel in ((2, "k_generate"), (8, "k_generate_pipe_multi"), (3, "k_generate")):
        ra, rb = rings(a, T, "fla", None), rings(b, T, "fla", None)
        a.generate(T, *ra)
        b.generate(T, *rb)
        torch.cuda.synchronize()
        assert a.generate_kernel() == kernel and b.generate_kernel() == "k_generate"
        for x, y in zip(ra, rb):
            assert same(x, y), T
        del ra, rb
    sa, sb = a.get_state(), b.get_state()
    assert all(same(sa[k], sb[k]) for k in sa)
    a.close()
    b.close()


def test_pipelined_rings_against_the_oracle(goenv, monkeypatch):
    n, T, seed = 40, 30, 77
    env = make(goenv, monkeypatch, True, n, seed, max_ply=70, pace=3000)
    dev = env.device
    augs = (torch.arange(n, dtype=torch.int32, device=dev) % 8).contiguous()
    oracles = [OracleBoard() for _ in range(n)]
    ended = 0
    for launch in range(4):
        feat, legal, acts = rings(env, T, "fla", augs)
        env.generate(T, feat, legal, acts, aug_codes=augs)
        f, lg, ac = feat.cpu().numpy(), legal.cpu().numpy(), acts.cpu().numpy()
        for b in range(n):
            o = oracles[b]
            for t in range(T):
                if o.is_game_end() or o.canonical()["scalars"][0] > 70:     # the device restarts it before this ply
                    o = oracles[b] = OracleBoard()
                    ended += 1
                assert o.forward(int(ac[t, b]))[0], (launch, t, b)
                assert np.array_equal(lg[t, b], o.legal()), (launch, t, b)
                assert_planes_equal(f[t, b], o.extract(b % 8), "launch %d ply %d board %d" % (launch, t, b))
    assert ended >= n
    env.close()


def test_preset_planes_and_graph_capture(goenv, monkeypatch):
    """The constant planes preset (the eleven planes are the caller's) through the pipelined kernel, and a launch
    recorded into a HIP graph: the replays continue the games exactly as plain calls do."""
    n, T, seed = 128, 8, 9
    plain = make(goenv, monkeypatch, True, n, seed)
    graphed = make(goenv, monkeypatch, True, n, seed)
    for e in (plain, graphed):
        e.set_constant_planes_preset(True)
    dev = plain.device
    rp, rg = rings(plain, T, "fla", None), rings(graphed, T, "fla", None)
    for r in (rp, rg):
        r[0].zero_()
        r[0][:, :, 12] = 3.5                               # a sentinel in a constant plane must survive
    plain.generate(T, *rp)
    graphed.generate(T, *rg)                               # one call outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        graphed.generate(T, *rg)
    for _ in range(5):
        plain.generate(T, *rp)
        g.replay()
        torch.cuda.synchronize()
        for x, y in zip(rp, rg):
            assert same(x, y)
        assert float(rg[0][:, :, 12].min()) == 3.5 and float(rg[0][:, :, 16:].abs().max()) == 0.0
    sp, sg = plain.get_state(), graphed.get_state()
    for k in sp:
        assert same(sp[k], sg[k]), k
    plain.close()
    graphed.close()


def test_rate_limiter_tunes_itself_and_changes_no_byte(goenv, monkeypatch):
    """Default settings at a size where the limiter matters: forty launches, during which the tuner walks through its
    candidate paces; every ring equals the two-wave kernel's, and the tuner reports a pace and its estimates."""
    n, T, seed = 1024, 16, 3
    monkeypatch.delenv("GOENV_GEN_PIPE", raising=False)
    monkeypatch.delenv("GOENV_GEN_PACE_NS", raising=False)
    tuned = goenv.GoEnv(n, seed=seed, max_ply=722)
    ref = make(goenv, monkeypatch, False, n, seed, max_ply=722)
    rt, rr = rings(tuned, T, "fla", None), rings(ref, T, "fla", None)
    for launch in range(40):
        tuned.generate(T, *rt)
        ref.generate(T, *rr)
        if launch % 5 == 4:
            torch.cuda.synchronize()
            for x, y in zip(rt, rr):
                assert same(x, y), launch
    torch.cuda.synchronize()
    pace, est = tuned.generate_pace()
    assert 6.0 <= pace <= 7.5, pace
    assert sum(1 for e in est if e > 0) >= 8, est
    assert ref.generate_pace()[0] == 0.0
    tuned.close()
    ref.close()


def test_first_call_inside_a_capture(goenv, monkeypatch):
    """A context whose very first goenv_generate is recorded into a graph: the tuner records no event there and uses its safe
    pace; the replays continue the games exactly as plain calls on a twin context do."""
    n, T, seed = 1024, 6, 21
    monkeypatch.delenv("GOENV_GEN_PIPE", raising=False)
    monkeypatch.delenv("GOENV_GEN_PACE_NS", raising=False)
    plain = goenv.GoEnv(n, seed=seed, max_ply=722)
    cold = goenv.GoEnv(n, seed=seed, max_ply=722)
    rp, rc = rings(plain, T, "fla", None), rings(cold, T, "fla", None)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cold.generate(T, *rc)
    for _ in range(4):
        plain.generate(T, *rp)
        g.replay()
        torch.cuda.synchronize()
        for x, y in zip(rp, rc):
            assert same(x, y)
    assert cold.generate_kernel() == "k_generate_pipe" and cold.generate_pace()[0] == 0.0   # nothing was timed
    plain.close()
    cold.close()
