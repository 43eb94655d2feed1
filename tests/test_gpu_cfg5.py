"""cfg 5 (BASELINE.json configs[4]) as a test: the df_selfplay loop -- go_loader in selfplay mode, a
small df_model2-shaped PolicyValueNet on the GPU as the `actor` callback, GCWrapper.Run -- with the
sequencing invariants of check.py:33-91 and every sampled move checked for legality against the
`legal` key of the batch it was sampled for.  Also the 2-rank form of the selfplay tool (ranks share
the card over gloo): different ranks play different games, counts add up."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_df_selfplay_loop_small_net():
    from elf_amd import go_loader
    from elf_amd.df_model import PolicyValueNet, make_actor
    N = 64
    args = go_loader.add_arguments().parse_args(
        ["--mode", "selfplay", "--num_games", str(N), "--batchsize", "32", "--num_collectors", "0", "--gpu", "0",
         "--seed", "3", "--legal_mask", "--ply_cap", "40", "--end_on_two_passes", "--signal_restart",
         "--additional_labels", "id,seq,game_counter,last_terminal,move_idx"])
    GC = go_loader.initialize(args)
    assert GC.params["num_action"] == 361 and GC.params["num_planes"] == 25
    torch.manual_seed(0)
    net = PolicyValueNet(dim=32, num_block=2).cuda().eval()
    model = make_actor(net)
    last = {}          # game -> (color planes before the move, legal mask, move_idx, seq, game_counter)
    seen = dict(batches=0, terminals=0, moves=0)

    def actor(batch):
        b = batch.hist(0)
        s, legal = b["s"], b["legal"]
        assert s.is_cuda and tuple(s.shape) == (batch.batchsize, 25, 19, 19)
        ids = b["id"].cpu().tolist()
        assert len(set(ids)) == len(ids)                              # ids unique within a batch
        occ = (s[:, 7] + s[:, 8]).reshape(-1, 361).cpu()              # our + opponent stones
        lg, seq = legal.cpu(), b["seq"].cpu().tolist()
        gcs, lt, mi = b["game_counter"].cpu().tolist(), b["last_terminal"].cpu().tolist(), b["move_idx"].cpu().tolist()
        for r, g in enumerate(ids):
            if g in last:
                p_occ, p_legal, p_mi, p_seq, p_gc = last[g]
                if lt[r]:                                             # the game was scored and restarted
                    seen["terminals"] += 1
                    assert seq[r] == 0 and gcs[r] == p_gc + 1 and mi[r] == 1 and int(occ[r].sum()) == 0
                else:
                    assert seq[r] == p_seq + 1 and gcs[r] == p_gc and mi[r] == p_mi + 1
                    new = ((occ[r] > 0) & (p_occ == 0)).nonzero().flatten().tolist()
                    assert len(new) <= 1                              # one stone appeared (none: a pass)
                    for p in new:
                        assert p_legal[p] == 1, "game %d played the illegal point %d" % (g, p)
                    seen["moves"] += 1
            last[g] = (occ[r].clone(), lg[r].clone(), mi[r], seq[r], gcs[r])
        seen["batches"] += 1
        out = model(batch)
        assert tuple(out["pi"].shape) == (batch.batchsize, 361) and tuple(out["V"].shape) == (batch.batchsize,)
        assert torch.allclose(out["pi"].sum(1), torch.ones(batch.batchsize, device=s.device), atol=1e-4)
        return out

    GC.reg_callback("actor", actor)
    GC.Start()
    for _ in range(2 * 50):                                           # every game passes the ply cap once
        GC.Run()
    stats = GC.GC.Stats()
    GC.Stop()
    assert seen["batches"] == 100 and seen["moves"] > 2000 and seen["terminals"] >= N // 2
    assert stats[0] >= seen["terminals"]


def test_selfplay_tool_two_ranks_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    common = ["--num_games", "64", "--batchsize", "64", "--iters", "45", "--num_block", "1", "--dim", "16",
              "--ply_cap", "30", "--json"]
    env = dict(os.environ, GOENV_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tool = os.path.join(ROOT, "tools", "df_selfplay_gpu.py")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), tool] + common,
                        capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r2.returncode == 0, r2.stderr[-2000:]
    r1 = subprocess.run([sys.executable, tool] + common, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r1.returncode == 0, r1.stderr[-2000:]
    two = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    one = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    # 50 steps of every game against a ply cap of 30: each game ends at least once on every rank
    assert one["games_finished"] >= 64 and two["games_finished"] >= 2 * 64
