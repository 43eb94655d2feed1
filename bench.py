#!/usr/bin/env python3
"""bench.py -- 19x19 Go moves/s of the MI355X batched Go environment (BASELINE.json metric).

The hot path over the whole batch resident in HBM: on-device cfg-2 policy (uniform over legal moves
that do not fill an own true eye) -> rules update (TryPlay/Play) -> legal mask -> 25 feature planes,
all written into device batch buffers.  Finished games are scored, counted into the stats vector and
restarted in place (as ELF's selfplay restarts games, go/game.cc:81-90), so every ply applies exactly
one move on every board.

A STEP is one pass of the timed kernel over one batch of output: ONE launch of the persistent
generator (goenv_generate, kernel k_generate_pipe), which plays `plies_per_step` plies (default 128, the
ring depth) of every board with the boards kept in LDS/registers and writes ply t's planes, mask and
move to slot t of ring buffers -- nothing is skipped against a launch per ply (same bytes, same
order; tests/test_gpu_entry_points.py holds the two paths bit-identical).  So
    value = n_gpus * boards * plies_per_step * steps / time      [moves/s]
and --steps K times exactly K launches.  (Round 1 called a single ply a step; the driver's K = 20 was
then ONE 0.18-ms launch, a fifth of it launch + sync.)  The launch-per-ply path (goenv_ply) and the
GameContext API path (what a model in the loop uses) are timed beside it under "extra".

Workload at N=1: BASELINE.json configs[1] -- 1,024 concurrent games, uniform-random legal-move
playouts to the end, 1 MI355X.  For N>1 every rank runs the same per-GPU batch (weak scaling),
boards keyed by global id; the only collective is the end-of-run stats all-reduce (RCCL).

Hygiene of the timed region: the boards are first rolled --preroll plies forward WITHOUT outputs
(set-up, not warmup) so that they stand in the steady state of continuous selfplay -- games of every
age, some ending, being scored and restarted inside any window; every warmup step is a full launch
of the timed shape over the WHOLE ring, so no ring slot is first touched inside the timed region;
the K-step region (barrier + synchronize on both sides) is timed `reps` times back to back and the
MEDIAN region is reported (ms_per_step x steps = one region).

    python bench.py --gpus N --steps K --warmup W [--boards B] [--ring T] [--preroll P]
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALG_BYTES_PLANES = 25 * 361 * 4          # float32 planes written per board per ply
ALG_BYTES_LEGAL = 361                    # u8 legal mask written per board per ply
REC_BYTES = 1632                         # packed board record: read once + written once per ply
ALG_BYTES_MOVE = 4                       # int32 move written per board per ply (generator)
ALG_BYTES_PER_BOARD = ALG_BYTES_PLANES + ALG_BYTES_LEGAL + 2 * REC_BYTES      # launch per ply
ALG_BYTES_PER_BOARD_PLY = ALG_BYTES_PLANES + ALG_BYTES_LEGAL + ALG_BYTES_MOVE  # generator, per ply
HBM_PEAK_GBS = 8000.0                    # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_share():
    """(cores this process may use, logical CPUs of the host).  The share is the smallest of the
    scheduler affinity, the cgroup CPU quota and the pool size the box exports (OMP_NUM_THREADS):
    on the GPU box os.cpu_count() shows the whole machine, the job owns 16 of its CPUs."""
    host = os.cpu_count() or 1
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            share = min(share, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    try:
        share = min(share, max(1, int(os.environ["OMP_NUM_THREADS"])))
    except (KeyError, ValueError):
        pass
    return max(1, share), host


def cpu_baseline(budget_sec=12.0):
    """The reference's own board.cc/board_feature.cc (oracle/_ref, kind "reference") -- or, when
    that build did not travel, our C port of it (kind "port") -- driving the same cfg-2 workload
    with Extract on every ply, on every core this job owns.  Bounded sample."""
    from oracle.pyoracle import OracleBoard, RefBoard
    threads, host = cpu_share()
    cls, kind = (RefBoard, "reference") if RefBoard.available() else (OracleBoard, "port")
    n_boards = threads * 4      # each thread plays games back to back until the budget is spent
    plies, sec = cls.bench_playouts(n_boards, threads, 1, 722, True, 1000000, budget_sec)
    plies_nf, sec_nf = cls.bench_playouts(n_boards, threads, 1, 722, False, 1000000, budget_sec / 3)
    return {
        "value": plies / sec, "unit": "moves/s", "cores": threads, "threads": threads, "host_logical_cpus": host,
        "kind": kind,
        "sample": "cfg-2 policy playouts with BoardFeature::Extract on every ply, one thread per core of this "
                  "job's CPU share (%d of the host's %d logical CPUs), %d plies in %.1f s (budget %.0f s)"
                  % (threads, host, plies, sec, budget_sec),
        "value_without_features": plies_nf / sec_nf,
        # ELF's own structure for the same job (not the favourable thread pool above): the unmodified reference
        # go_game module in selfplay mode, ONE OS THREAD PER GAME, Extract + collector hand-shake + forward per move,
        # numpy batch buffers, a reply that costs nothing (tools/time_selfplay_ref.py)
        "elf_thread_per_game": elf_selfplay_reference(),
    }


def elf_selfplay_reference():
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_selfplay_ref.py"), "--games", "1024",
                            "--batch", "1024", "--batches", "60"], capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}


def measured_traffic(boards, plies_per_launch):
    """HBM bytes per launch of the generator kernel FROM THE COMMITTED PROFILE (not measured in this
    run): FETCH_SIZE + WRITE_SIZE PMC passes over this same command, corrected as
    MI355X_MICROARCH.md prescribes (tools/profile_round.sh -> profiles/rNN_pmc.json), scaled to this
    run's plies per launch (the kernel writes the same bytes for every ply).  (None, None) when no
    pass exists for this batch size."""
    for name in ("r03_pmc.json", "r02_pmc.json", "r01_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if int(d["boards"]) != boards or "k_generate" not in d["kernel"]:
                continue
            return float(d["traffic_bytes"]) * plies_per_launch / float(d["plies_per_launch"]), "profiles/" + name
        except (OSError, ValueError, KeyError):
            continue
    return None, None


def plies_per_step(boards, ring, budget_bytes=6e9):
    """Plies one generator launch plays = ring depth, cut down for big batches so that the ring of
    planes stays within a few GB (8,192 boards: 20 plies; 65,536: 2)."""
    return int(max(1, min(ring, budget_bytes // (boards * ALG_BYTES_PER_BOARD_PLY))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32, help="timed generator launches (one step = --ring plies of every board)")
    ap.add_argument("--warmup", type=int, default=4, help="untimed launches of the same shape before the timed region")
    ap.add_argument("--boards", type=int, default=1024, help="boards per GPU (cfg 2: 1024; cfg 4 shard: 8192)")
    ap.add_argument("--ring", type=int, default=128,
                    help="ring depth in plies = plies per step (per generator launch); 4.8 GB of planes at 1,024 boards "
                         "(64 until round 3: a launch boundary costs the pipelined generator about 28 us)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--preroll", type=int, default=4096,
                    help="plies every board is rolled forward without outputs before the warmup (set-up: steady state)")
    ap.add_argument("--reps", type=int, default=0, help="timed regions (0 = as many as fit about 0.3 s, 1..15, odd)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()
    if args.steps < 1 or args.warmup < 0 or args.ring < 1 or args.boards < 1 or args.preroll < 0:
        ap.error("--steps, --ring and --boards must be positive, --warmup non-negative")

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # GOENV_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks
    # share the cards round-robin); the driver's runs use the default, nccl = RCCL, one rank per GPU
    backend = os.environ.get("GOENV_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    # GOENV_BENCH_FORCE_DIST=1: the process group (RCCL init with device_id, the two stats all-reduces, the
    # barriers, the MAX over ranks of the region time, destroy) also at world size 1 -- every collective call
    # of the N>1 path on the one GPU a test box has (tests/test_gpu_multirank.py)
    dist_on = world > 1 or os.environ.get("GOENV_BENCH_FORCE_DIST") == "1"
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run "
                  "--nproc-per-node %d" % (args.gpus, world, args.gpus), file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)

    from elf_amd import goenv
    from elf_amd.sharding import reduce_stats

    torch.cuda.set_device(local_rank)
    B = args.boards
    env = goenv.GoEnv(B, device=local_rank, seed=args.seed, board_id_base=rank * B, komi=7.5, max_ply=722)
    T = plies_per_step(B, args.ring)
    feat = torch.empty(T, B, 25, 19, 19, dtype=torch.float32, device=env.device)   # ring: slot t = ply t of a launch
    legal = torch.empty(T, B, 361, dtype=torch.uint8, device=env.device)
    moves = torch.empty(T, B, dtype=torch.int32, device=env.device)

    def steps(k):
        # k steps = k launches of the persistent generator, T plies of every board each, queued back to back
        for _ in range(k):
            env.generate(T, feat, legal, moves)

    def sync():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # set-up: into the steady state of continuous selfplay, statistics restarted.  No outputs, and through the
    # launch-per-ply kernel (same positions): a profile of this command then shows k_generate with full
    # launches of the timed shape only
    env.ply_repeat(args.preroll, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    env.stats_clear()
    # warmup: W full launches of the timed shape (each covers the whole ring); never fewer than one
    warm_steps = max(args.warmup, 1)
    steps(warm_steps)
    launches = args.steps

    def region():
        sync()
        env.timer_begin()
        t0 = time.perf_counter()
        steps(args.steps)
        env.timer_end()
        env.timer_spin()                              # poll, do not sleep, until the last launch is done
        torch.cuda.synchronize()
        ms = env.timer_ms() / launches                # HIP events on the launch stream; average launch
        sync()
        dt = time.perf_counter() - t0
        if dist_on:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ms

    regions = [region()]
    reps = args.reps if args.reps > 0 else max(1, min(15, int(0.3 / max(regions[0][0], 1e-6))))
    reps |= 1                                          # odd: the median is one measured region
    while len(regions) < reps:
        regions.append(region())
    regions.sort()
    elapsed, kernel_ms = regions[len(regions) // 2]
    kernel_ms = sorted(r[1] for r in regions)[len(regions) // 2]

    gen_kernel = env.generate_kernel()
    gen_pace, gen_est = env.generate_pace()
    # end-of-episode statistics: count the games still in flight, then the one collective
    env.finish()
    stats = reduce_stats(env.stats_tensor()).cpu().tolist()

    value = world * B * T * args.steps / elapsed
    alg_launch = B * (T * ALG_BYTES_PER_BOARD_PLY + 2 * REC_BYTES)
    achieved = alg_launch / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(B, T)
    out = {
        "metric": "19x19 Go moves/sec (node, 1/2/4/8 GPU) + feature-extract HBM GB/s",
        "value": value, "unit": "moves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "moves_per_step": world * B * T,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "i32", "data": "synthetic",
        "config": {"workload": "cfg2: %d concurrent 19x19 games per GPU, on-device uniform-random legal-move "
                               "playouts to the end (scored, counted, restarted in place); every ply of every "
                               "board = rules update + legal mask + 25 planes + the move, written to slot t of "
                               "device ring buffers by the persistent generator (goenv_generate); one step = one "
                               "launch = %d plies of every board" % (B, T),
                   "boards_per_gpu": B, "global_boards": world * B, "ply_cap": 722,
                   "plies_per_step": T, "moves_per_step_per_gpu": B * T, "launches": launches,
                   "parallelism": "shard%d" % world,
                   "collective_backend": backend if dist_on else None,
                   "preroll_plies": args.preroll, "warmup_steps_run": warm_steps, "reps": len(regions),
                   "region_ms": [round(r[0] * 1e3, 4) for r in regions], "reported": "median region"},
        "roofline": {"bound": "hbm", "kernel": gen_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "frac_wall": alg_launch * launches / elapsed / 1e9 / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_unit": "bytes per launch, from %s (rocprofv3 PMC passes of this "
                                                         "command; not measured in this run)" % traffic_src,
                     "algorithmic_bytes_per_launch": alg_launch,
                     "bytes_per_board_ply": ALG_BYTES_PER_BOARD_PLY, "record_bytes_per_board_launch": 2 * REC_BYTES,
                     "kernel_ms": kernel_ms,
                     # the generator's rate limiter tunes itself (include/goenv.h, goenv_generate_pace): the pace it
                     # settled on (ns offered per board-ply) and its time estimates (us per ply) per candidate pace
                     "store_pace_ns_per_board_ply": gen_pace, "pace_estimates_us_per_ply": [round(e, 3) for e in gen_est]},
        "feature_extract_GBps": achieved,
        "stats": {"games": stats[0], "black_wins": stats[1], "white_wins": stats[2], "plies": stats[4],
                  "max_len": stats[9], "ply_cap_hits": stats[8]},
    }
    env.close()
    del feat, legal, moves
    torch.cuda.empty_cache()

    if rank == 0 and world == 1 and not args.no_extra:
        out["extra"] = extra_measurements(goenv, args.seed)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


def extra_measurements(goenv, seed):
    """Not the headline: the launch-per-ply path (k_ply, what a model in the loop uses) at 1,024
    boards, at the cfg-4 per-GPU shard (8,192 boards) and at 65,536 boards, the standalone feature
    kernel, the generator at the cfg-4 shard, and the persistent playout kernel."""
    res = {}
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    for B in (1024, 8192, 65536):
        env = goenv.GoEnv(B, seed=seed, max_ply=722)
        feat, legal = env.new(torch.float32, 25, 19, 19), env.new(torch.uint8, 361)
        steps = {1024: 2000, 8192: 300, 65536: 60}[B]
        env.ply_repeat(150 if B == 65536 else 100, flags, feat_out=feat, legal_out=legal)   # into the middle game
        torch.cuda.synchronize()
        env.timer_begin()
        env.ply_repeat(steps, flags, feat_out=feat, legal_out=legal)
        env.timer_end()
        ms = env.timer_ms() / steps
        r = {"ply_ms": ms, "moves_per_s": B / (ms * 1e-3),
             "ply_GBps": B * ALG_BYTES_PER_BOARD / (ms * 1e-3) / 1e9}
        steps = min(steps, 300)
        env.timer_begin()
        for _ in range(steps):
            env.features(feat, None, legal)
        env.timer_end()
        ms = env.timer_ms() / steps
        r["features_ms"] = ms
        r["features_GBps"] = B * (ALG_BYTES_PLANES + ALG_BYTES_LEGAL + REC_BYTES) / (ms * 1e-3) / 1e9
        res["launch_per_ply_%d" % B] = r
        env.close()
        del feat, legal
    # generator at the cfg-4 shard: 16 plies per launch (4.7 GB of planes per launch); and over 65,536 boards, 8 plies per
    # launch (19 GB): more boards than one launch of the pipelined kernel holds, cut into chunks of 8,192
    for B, T in ((8192, 16), (65536, 8)):
        env = goenv.GoEnv(B, seed=seed, max_ply=722)
        fr = torch.empty(T, B, 25, 19, 19, device=env.device)
        lr = torch.empty(T, B, 361, dtype=torch.uint8, device=env.device)
        ar = torch.empty(T, B, dtype=torch.int32, device=env.device)
        for _ in range(2 * goenv.GEN_PACE_CANDIDATES + 6):   # the generator's rate limiter walks through its candidate paces first
            env.generate(T, fr, lr, ar)
        torch.cuda.synchronize()
        env.timer_begin()
        for _ in range(10):
            env.generate(T, fr, lr, ar)
        env.timer_end()
        ms = env.timer_ms() / (10 * T)
        res["generate_%d" % B] = {"ply_ms": ms, "moves_per_s": B / (ms * 1e-3), "plies_per_launch": T,
                                  "GBps": B * ALG_BYTES_PER_BOARD_PLY / (ms * 1e-3) / 1e9, "kernel": env.generate_kernel(),
                                  "store_pace_ns_per_board_ply": env.generate_pace()[0]}
        env.close()
        del fr, lr, ar
        torch.cuda.empty_cache()
    # the API path the north star names: GameContext selfplay, Wait -> reply pi on the device -> Steps
    for B in (1024, 8192):
        res["game_context_selfplay_%d" % B] = game_context_selfplay(B, seed, True)
        res["game_context_selfplay_%d" % B]["vs_launch_per_ply"] = (
            res["game_context_selfplay_%d" % B]["moves_per_s"] / res["launch_per_ply_%d" % B]["moves_per_s"])
        res["game_context_selfplay_%d" % B]["unfused_round1_path"] = game_context_selfplay(B, seed, False, steps=300)
        # optional mode, never the headline: tensors zero-filled once, the 11 constant planes (44 % of the bytes) never
        # written again (GameOptions.preset_zero_planes); planes stay bit-identical (tests/test_gpu_soak.py)
        res["game_context_selfplay_%d" % B]["preset_zero_planes"] = game_context_selfplay(B, seed, True, preset=True)
    res["cfg3_offline_256"] = cfg3_offline()
    res["mcts_256x16x100"] = mcts_search()
    res["cfg5_selfplay_fp32"] = cfg5_selfplay(["--num_games", "1024", "--batchsize", "1024", "--iters", "15"])
    res["cfg5_selfplay_bf16"] = cfg5_selfplay(["--num_games", "2048", "--batchsize", "2048", "--iters", "15", "--bf16"])
    # the rules-only step kernel (k_playout: whole games with the board in LDS/registers, no planes) -- the
    # LDS/VALU-bound half of the path; its counters against gfx950 peaks: profiles/r03_playout.md, r03b_playout.md
    for B in (1024, 8192, 65536):
        env = goenv.GoEnv(B, seed=seed, max_ply=722)
        nm = env.new(torch.int32)
        env.playout(None, nm)
        torch.cuda.synchronize()
        env.reset()
        env.timer_begin()
        env.playout(None, nm)
        env.timer_end()
        ms = env.timer_ms()
        res["persistent_playout_%d" % B] = {"ms": ms, "moves": int(nm.sum().item()),
                                            "moves_per_s": float(nm.sum().item()) / (ms * 1e-3)}
        env.close()
    return res


def game_context_selfplay(B, seed, fuse, steps=1500, warm=100, preset=False):
    """GameContext in selfplay mode (elf_amd.go_game, the drop-in for the reference's pybind module):
    B games, ONE collector of batch B, device tensors `s` + `legal` in, `pi` + `V` out; the reply is a
    constant uniform pi already on the device (no model), so the number is the environment's own
    API-level rate: Wait -> Steps, every Steps = sample a legal move from pi + rules + the next
    batch's mask and 25 planes.  host_us_per_batch = time spent inside Wait() + Steps() themselves."""
    from elf_amd import go_game as go
    co = go.ContextOptions()
    co.num_games = B
    opt = go.GameOptions()
    opt.mode, opt.seed, opt.ply_cap, opt.end_on_two_passes, opt.fuse_selfplay = "selfplay", seed, 722, True, fuse
    opt.preset_zero_planes = preset
    gc = go.GameContext(co, opt)
    gs = gc.CreateGroupStat()
    gs.name = "actor"
    gid = gc.AddCollectors(B, 0, 0, gs)
    tt = {"float": torch.float32, "unsigned char": torch.uint8}
    keep = []
    for which, keys in (("input", ("s", "legal")), ("reply", ("pi", "V"))):
        for key in keys:
            info = gc.GetTensorSpec(gid, key, 1)
            v = (torch.zeros if preset and key == "s" else torch.ones)(*info.sz, dtype=tt[info.type], device="cuda:0")
            info.p, info.byte_size = v.data_ptr(), v.numel() * v.element_size()
            gc.AddTensor(gid, which, info)
            keep.append(v)
    gc.Start()
    fuse_max = gc.FuseMax()
    host = 0.0
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            t0, host = time.perf_counter(), 0.0
        h0 = time.perf_counter()
        infos = gc.Wait(0)
        gc.Steps(infos)
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = gc.Stats()
    gc.Stop()
    return {"moves_per_s": B * steps / dt, "step_ms": dt * 1e3 / steps, "host_us_per_batch": host * 1e6 / steps,
            "launches_per_step": (1 if B <= fuse_max else 2) if fuse else 3, "fuse_max": fuse_max, "steps": steps, "games_finished": stats[0],
            "GBps": B * (ALG_BYTES_PER_BOARD - (11 * 361 * 4 if preset else 0)) / (dt / steps) / 1e9,
            "plane_bytes_written_per_board": ALG_BYTES_PLANES - (11 * 361 * 4 if preset else 0)}


def cfg5_selfplay(flags):
    """BASELINE.json configs[4] on one GPU: the df_selfplay loop with a df_model2-shaped 20-block x 128-channel
    PyTorch-ROCm net, random weights, moves sampled on the device from pi (tools/df_selfplay_gpu.py in a child
    process; the 8-GPU form is the same tool under torch.distributed.run)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "df_selfplay_gpu.py"), "--json"] + flags,
                           capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}


def mcts_search():
    """tree search behind the GameContext, 256 games x 16 search threads x 100 rollouts, leaf
    evaluations per second with a reply that costs nothing (tools/time_mcts.py in a child process; beside it
    the UNMODIFIED reference module, 16 games x 16 threads = 256 OS threads on this host's cores)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_mcts.py"), "--evals", "150000"],
                           capture_output=True, text=True, timeout=400)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}


def cfg3_offline(games=256, batches=150):
    """BASELINE.json configs[2]: offline SGF replay, batch 256 (tools/time_offline.py in a child process:
    ours with device tensors, and the UNMODIFIED reference go_game module built under oracle/_ref when
    it travelled -- one OS thread per game, numpy buffers -- on this host's cores)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_offline.py"), "--games", str(games),
                            "--batches", str(batches)], capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else {"error": r.stderr[-300:]}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"error": repr(e)}


if __name__ == "__main__":
    main()
