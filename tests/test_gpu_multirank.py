"""GPU tests of the N>1 path as far as one GPU allows.

* goenv_stats_allreduce (the C-ABI collective, RCCL) with a one-rank communicator built through
  librccl's own C API: the call runs on the context's stream and leaves a one-rank sum unchanged.
* bench.py's N>1 path with two ranks sharing the card (GOENV_BENCH_BACKEND=gloo: RCCL refuses two
  ranks on one device): launched as the driver launches it, its reduced statistics must equal those
  of ONE process holding both ranks' boards -- partition invariance + the all-reduce, end to end."""
import ctypes as C
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stats_allreduce_through_the_c_abi_one_rank():
    from elf_amd import goenv
    goenv.load_library()
    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    torch.cuda.set_device(0)
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    env = goenv.GoEnv(64, seed=3, max_ply=30)
    env.ply_repeat(100, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    before = env.stats()
    assert before[goenv.ST_GAMES] > 0 and before[goenv.ST_MAX_LEN] > 0
    env.stats_allreduce(comm.value)
    assert env.stats() == before                      # one rank: SUM and MAX are the identity
    env.stats_allreduce(comm.value)
    assert env.stats() == before
    with pytest.raises(goenv.GoEnvError):
        env.stats_allreduce(0)
    env.close()
    rccl.ncclCommDestroy(comm)


def _bench(args, ranks):
    env = dict(os.environ, GOENV_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "3", "--warmup", "1", "--ring", "16", "--preroll", "430", "--reps", "1", "--no-extra",
              "--no-cpu-baseline"] + args
    if ranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common
    else:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
               "--gpus", str(ranks)] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 prints ONE line
    return json.loads(lines[0])


def test_bench_two_ranks_reduce_to_the_stats_of_one_context():
    two = _bench(["--boards", "256"], 2)
    one = _bench(["--boards", "512"], 1)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and two["config"]["global_boards"] == 512
    assert two["config"]["collective_backend"] == "gloo"
    assert one["stats"]["games"] > 0
    assert two["stats"] == one["stats"]                  # rank r plays global boards [256 r, 256 r + 256)
    # whole-job value = all ranks' moves / max-over-ranks time
    moves = two["config"]["global_boards"] * two["config"]["plies_per_step"] * two["steps"]
    assert abs(two["value"] - moves / (two["ms_per_step"] * 1e-3 * two["steps"])) < 1e-6 * two["value"]


def _launch_one_rank(script_args, force_nccl):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("GOENV_BENCH_BACKEND", None)                      # default backend: nccl = RCCL
    if force_nccl:
        env["GOENV_BENCH_FORCE_DIST"] = "1"
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
               "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_args
    else:
        cmd = [sys.executable] + script_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_rccl_with_one_rank_equals_the_plain_run():
    """The driver's N>1 launch line with ONE rank over RCCL (GOENV_BENCH_FORCE_DIST=1): process-group init with
    device_id, dist.barrier(), the int64 SUM and MAX all-reduces of reduce_stats, the MAX over ranks of the
    region time, destroy_process_group -- every collective call of the N>1 path, on the one GPU this box has.
    Its statistics must equal the undistributed run's."""
    common = ["--gpus", "1", "--steps", "3", "--warmup", "1", "--ring", "16", "--preroll", "430", "--reps", "1",
              "--no-extra", "--no-cpu-baseline", "--boards", "256"]
    script = [os.path.join(ROOT, "bench.py")] + common
    rccl = _launch_one_rank(script, True)
    plain = _launch_one_rank(script, False)
    assert rccl["config"]["collective_backend"] == "nccl" and plain["config"]["collective_backend"] is None
    assert rccl["n_gpus"] == 1 and rccl["stats"] == plain["stats"] and rccl["stats"]["games"] > 0
    assert rccl["config"]["moves_per_step_per_gpu"] == plain["config"]["moves_per_step_per_gpu"]


def test_selfplay_tool_under_rccl_with_one_rank():
    """tools/df_selfplay_gpu.py (cfg 5) the same way: RCCL process group first, then the GPU; barrier, stats
    all-reduce (SUM + MAX) and the float64 SUM / MAX of the timing vector all run over RCCL."""
    script = [os.path.join(ROOT, "tools", "df_selfplay_gpu.py"), "--json", "--num_games", "64", "--batchsize", "64",
              "--iters", "6", "--num_block", "2", "--dim", "32"]
    rccl = _launch_one_rank(script, True)
    plain = _launch_one_rank(script, False)
    assert rccl["n_gpus"] == 1 and rccl["moves_per_s"] > 0
    assert rccl["games_finished"] == plain["games_finished"]
