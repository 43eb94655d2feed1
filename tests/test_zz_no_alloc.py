"""goenv_features_static -- the entry point that allocates nothing (include/goenv.h): its planes and
legal masks must equal, bit for bit, those of the existing goenv_set_state + goenv_features on a
context, on two calls in a row, the second with a smaller input (the static scratch array keeps the
first call's records).  Also checked against the CPU oracle, and that free device memory does not
move across a call."""
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


def positions(goenv, n, seed, plies):
    """n mid-game positions (canonical view, device tensors) from the on-device policy."""
    env = goenv.GoEnv(n, seed=seed, max_ply=722)
    env.ply_repeat(plies, goenv.F_RANDOM_POLICY)
    st = env.get_state()
    env.close()
    return st["color"], st["last_placed"], st["scalars"]


def via_context(goenv, color, lastp, scal, augs):
    n = color.shape[0]
    env = goenv.GoEnv(n, seed=1)
    env.set_state(color, lastp, scal)
    feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
    env.features(feat, augs, legal)
    torch.cuda.synchronize()
    env.close()
    return feat, legal


def test_static_entry_point_equals_context_path_twice(goenv):
    dev = torch.device("cuda", 0)
    for call, (n, seed, plies) in enumerate([(700, 41, 120), (93, 42, 37)]):   # the second call is smaller
        color, lastp, scal = positions(goenv, n, seed, plies)
        augs = ((torch.arange(n, dtype=torch.int32, device=dev) * 3 + call) % 8).contiguous()
        want_f, want_l = via_context(goenv, color, lastp, scal, augs)
        feat = torch.full((n, 25, 19, 19), -7.0, dtype=torch.float32, device=dev)
        legal = torch.full((n, 361), 9, dtype=torch.uint8, device=dev)
        if call == 1:
            free0 = torch.cuda.mem_get_info(dev)[0]
        goenv.features_static(color, lastp, scal, feat, legal, augs)
        torch.cuda.synchronize()
        if call == 1:
            assert torch.cuda.mem_get_info(dev)[0] == free0, "the call changed the free device memory"
        assert torch.equal(feat.view(torch.int32), want_f.view(torch.int32)), "planes differ on call %d" % call
        assert torch.equal(legal, want_l), "legal masks differ on call %d" % call
        # and against the oracle, on a sample of the rows
        c, lp, sc = color.cpu().numpy(), lastp.cpu().numpy(), scal.cpu().numpy()
        f, lg, ag = feat.cpu().numpy(), legal.cpu().numpy(), augs.cpu().tolist()
        for b in range(0, n, 17):
            o = OracleBoard()
            o.set_state(c[b], lp[b], sc[b])
            assert_planes_equal(f[b], o.extract(ag[b]), "call %d row %d" % (call, b))
            assert np.array_equal(lg[b], o.legal())


def test_static_entry_point_limits(goenv):
    dev = torch.device("cuda", 0)
    n = goenv.STATIC_MAX_BOARDS + 1
    color = torch.zeros(n, 361, dtype=torch.uint8, device=dev)
    scal = torch.zeros(n, 16, dtype=torch.int32, device=dev)
    legal = torch.zeros(n, 361, dtype=torch.uint8, device=dev)
    with pytest.raises(goenv.GoEnvError, match="GOENV_STATIC_MAX_BOARDS"):
        goenv.features_static(color, None, scal, None, legal)
    with pytest.raises(goenv.GoEnvError, match="no output buffer"):
        goenv.features_static(color[:4], None, scal[:4], None, None)
    # the maximum itself works; empty boards: every point legal, last_placed optional
    n = goenv.STATIC_MAX_BOARDS
    scal[:, 0] = 1
    scal[:, 1] = 1
    scal[:, 9] = -1
    goenv.features_static(color[:n], None, scal[:n], None, legal[:n])
    torch.cuda.synchronize()
    assert int(legal[:n].sum().item()) == n * 361
    assert int(legal[n:].sum().item()) == 0


def test_static_entry_point_inside_a_captured_graph(goenv):
    """What the entry point is for: a caller under HIP graph capture, where hipMalloc is not allowed.  One
    warm call (set-up), then the call is captured into a graph and the graph replayed on new positions."""
    dev = torch.device("cuda", 0)
    n = 257
    color, lastp, scal = positions(goenv, n, 51, 80)
    feat = torch.zeros(n, 25, 19, 19, device=dev)
    legal = torch.zeros(n, 361, dtype=torch.uint8, device=dev)
    goenv.features_static(color, lastp, scal, feat, legal)            # set-up happens here, outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        goenv.features_static(color, lastp, scal, feat, legal)
    for seed, plies in ((52, 33), (53, 150)):
        c2, l2, s2 = positions(goenv, n, seed, plies)
        color.copy_(c2)
        lastp.view(torch.int16).copy_(l2.view(torch.int16))
        scal.copy_(s2)
        feat.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        want_f, want_l = via_context(goenv, c2, l2, s2, None)
        assert torch.equal(feat.view(torch.int32), want_f.view(torch.int32))
        assert torch.equal(legal, want_l)


def test_static_entry_point_refuses_overlapping_calls(goenv):
    """All calls on one GPU share one array.  A call from another stream while the previous call's kernels
    are still in flight is refused with an error instead of racing; once they are done it goes through."""
    dev = torch.device("cuda", 0)
    n = 64
    color, lastp, scal = positions(goenv, n, 61, 60)
    feat = torch.zeros(n, 25, 19, 19, device=dev)
    legal = torch.zeros(n, 361, dtype=torch.uint8, device=dev)
    a, b = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        torch.cuda._sleep(400_000_000)                       # ~0.2 s of GPU time in front of the call on stream a
        goenv.features_static(color, lastp, scal, feat, legal)
        goenv.features_static(color, lastp, scal, feat, legal)   # same stream: ordered, allowed
    with torch.cuda.stream(b):
        with pytest.raises(goenv.GoEnvError, match="still in flight"):
            goenv.features_static(color, lastp, scal, feat, legal)
    torch.cuda.synchronize()
    with torch.cuda.stream(b):
        goenv.features_static(color, lastp, scal, feat, legal)
    torch.cuda.synchronize()
    want_f, want_l = via_context(goenv, color, lastp, scal, None)
    assert torch.equal(feat.view(torch.int32), want_f.view(torch.int32)) and torch.equal(legal, want_l)


def test_first_call_inside_a_capture_is_refused_with_a_clear_message():
    """The first call on a device uploads the history table with a synchronous copy.  Made inside a stream
    capture it must fail with a message that says so (fresh process: set-up happens once per process)."""
    import os
    import subprocess
    import sys
    code = (
        "import torch, sys\n"
        "sys.path.insert(0, %r)\n"
        "from elf_amd import goenv\n"
        "dev = torch.device('cuda', 0)\n"
        "c = torch.zeros(4, 361, dtype=torch.uint8, device=dev)\n"
        "s = torch.zeros(4, 16, dtype=torch.int32, device=dev); s[:, 0] = 1; s[:, 1] = 1; s[:, 9] = -1\n"
        "l = torch.zeros(4, 361, dtype=torch.uint8, device=dev)\n"
        "torch.cuda.synchronize()\n"
        "st = torch.cuda.Stream(dev)\n"
        "g = torch.cuda.CUDAGraph()\n"
        "try:\n"
        "    with torch.cuda.graph(g, stream=st):\n"
        "        goenv.features_static(c, None, s, None, l)\n"
        "    print('NO ERROR')\n"
        "except goenv.GoEnvError as e:\n"
        "    print('REFUSED:', e)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert "REFUSED:" in r.stdout and "before starting a stream capture" in r.stdout, (r.stdout[-400:], r.stderr[-400:])
