"""A whole selfplay step is capturable: goenv_ply_policy_sel -- what GameContext.Steps issues -- allocates
nothing (its one scratch buffer, the sampled moves of the two-launch form above 3,072 boards, is made by
goenv_create), so the step can be recorded into a HIP graph and replayed.  Both forms are captured here
(1,024 boards: one launch; 8,192 boards: sampler + ply) and every replay is compared with an uncaptured
context stepping the same boards with the same policy rows, and with the CPU oracle on a sample of boards.
Bar: bit-exact."""
import numpy as np
import pytest

from common import assert_canon_equal, assert_planes_equal
from oracle.pyoracle import OracleBoard

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("boards,steps", [(1024, 12), (8192, 6)])
def test_fused_selfplay_step_in_a_captured_graph(boards, steps):
    from elf_amd import goenv
    goenv.load_library()
    seed = 19
    env = goenv.GoEnv(boards, seed=seed, max_ply=722)       # stepped by graph replays
    twin = goenv.GoEnv(boards, seed=seed, max_ply=722)      # stepped by plain calls
    assert (boards > env.fuse_max()) == (boards == 8192)
    dev = env.device
    pi = torch.rand(boards, 361, device=dev)
    outs = [dict(feat=e.new(torch.float32, 25, 19, 19), legal=e.new(torch.uint8, 361), acts=e.new(torch.int32),
                 status=e.new(torch.int32)) for e in (env, twin)]
    flags = goenv.F_RESET_ILLEGAL | goenv.F_RESTART_AFTER_END

    def step(e, o):
        e.ply_policy_sel(pi, flags=flags, feat_out=o["feat"], legal_out=o["legal"], actions_out=o["acts"],
                         status_out=o["status"])

    step(env, outs[0])                                       # one warm step outside the capture, on both
    step(twin, outs[1])
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step(env, outs[0])
    # (capturing does not execute: the boards of `env` are still one step behind the graph's first replay)
    sample = list(range(0, boards, max(1, boards // 48)))
    oracles = {}
    st = {k: v.cpu().numpy() for k, v in twin.get_state().items()}
    for b in sample:
        o = OracleBoard()
        o.set_state(st["color"][b], st["last_placed"][b], st["scalars"][b])
        oracles[b] = o
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    for t in range(steps):
        pi.copy_(torch.rand(boards, 361, device=dev, generator=g))
        pi_h = pi.cpu().numpy()
        graph.replay()
        step(twin, outs[1])
        torch.cuda.synchronize()
        for k in outs[0]:
            a, b_ = outs[0][k], outs[1][k]
            assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a,
                               b_.view(torch.int32) if b_.dtype == torch.float32 else b_), (t, k)
        acts = outs[0]["acts"].cpu().numpy()
        f, lg = outs[0]["feat"].cpu().numpy().reshape(boards, -1), outs[0]["legal"].cpu().numpy()
        for b, o in oracles.items():
            want = o.sample_policy(seed, b, 0, pi_h[b], 0)     # no game ends within these few plies: game 0
            assert int(acts[b]) == want, (t, b)
            assert o.forward(want)[0]
            assert_planes_equal(f[b], o.extract(0), "step %d board %d" % (t, b))
            assert np.array_equal(lg[b], o.legal())
    sa = {k: v.cpu() for k, v in env.get_state().items()}
    sb = {k: v.cpu() for k, v in twin.get_state().items()}
    for k in sa:
        assert torch.equal(sa[k].view(torch.int16) if sa[k].dtype == torch.uint16 else sa[k],
                           sb[k].view(torch.int16) if sb[k].dtype == torch.uint16 else sb[k]), k
    assert torch.cuda.mem_get_info(dev)[0] >= free0 - (64 << 20), "the captured steps allocated device memory"
    env.close()
    twin.close()
