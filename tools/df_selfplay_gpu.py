#!/usr/bin/env python3
"""cfg 5: the df_selfplay loop (df_selfplay.py:12-54) with the GPU environment, on one GPU
or -- launched under torch.distributed.run -- on one rank per GPU.

The environment side is this repo's `go_game.GameContext` through `GCWrapper` (mode selfplay,
descriptor `actor`: input `s` (+`legal`), reply `pi`,`V` -- go/game.py:79-87); the model side is
plain PyTorch-ROCm (elf_amd/df_model.py, the shape of go/df_model2.py:28-94), random-initialised
(no checkpoints exist here).  Moves are sampled on the device from `pi` over the legal points.
Every rank plays its own `num_games` games (global ids rank * num_games + i key the RNG: weak
scaling, no data-path collective); at the end the statistics vector and the move counts are
all-reduced (RCCL) and rank 0 prints one line.

    python tools/df_selfplay_gpu.py --num_games 8192 --batchsize 2048 --iters 50
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 \\
        tools/df_selfplay_gpu.py --num_games 8192 --batchsize 2048 --iters 50 --bf16
"""
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("GOENV_BENCH_BACKEND", "nccl")
    if backend == "gloo":                     # rehearsal on a box with fewer GPUs than ranks
        local_rank %= max(1, torch.cuda.device_count())
    dist_on = world > 1 or os.environ.get("GOENV_BENCH_FORCE_DIST") == "1"   # (=1: the collectives at world size 1)
    if dist_on:                               # before anything touches the GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29532")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from elf_amd import go_loader
    from elf_amd.df_model import PolicyValueNet
    from elf_amd.sharding import reduce_stats

    ap = go_loader.add_arguments()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--num_block", type=int, default=20)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--json", action="store_true", help="rank 0 prints a JSON line instead of text")
    ap.set_defaults(mode="selfplay", num_games=4096, batchsize=1024, num_collectors=0, seed=1, ply_cap=722,
                    end_on_two_passes=True, signal_restart=True)
    args = ap.parse_args()
    args.gpu = local_rank if world > 1 or args.gpu is None else args.gpu
    args.board_id_base = rank * args.num_games
    torch.cuda.set_device(args.gpu)
    GC = go_loader.initialize(args)
    dev = torch.device("cuda", args.gpu)
    torch.manual_seed(1)                      # every rank builds the same random-initialised actor
    net = PolicyValueNet(dim=args.dim, num_block=args.num_block).to(dev).eval()
    if args.bf16:
        net = net.to(torch.bfloat16).to(memory_format=torch.channels_last)
    t_model = [0.0]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def actor(batch):
        ev0.record()
        with torch.no_grad():
            out = net(batch.hist(0))
        ev1.record()
        ev1.synchronize()
        t_model[0] += ev0.elapsed_time(ev1) * 1e-3
        return dict(pi=out["pi"].float(), V=out["V"].float().view(-1))

    GC.reg_callback("actor", actor)
    GC.Start()
    for _ in range(5):
        GC.Run()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    t_model[0] = 0.0
    t0 = time.perf_counter()
    moves = 0
    for _ in range(args.iters):
        GC.Run()
        moves += GC.infos.batchsize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = torch.tensor(GC.GC.Stats(), dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    tot = torch.tensor([float(moves), dt, t_model[0]], dtype=torch.float64, device=stats.device)
    if dist_on:
        stats = reduce_stats(stats)
        mx = tot[1:].clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        tot[1:] = mx
    moves_all, dt_all, model_all = (float(x) for x in tot.tolist())
    if rank == 0:
        res = dict(config="cfg5", n_gpus=world, games_per_gpu=args.num_games, batchsize=args.batchsize,
                   net="%d blocks x %d ch" % (args.num_block, args.dim), dtype="bf16" if args.bf16 else "fp32",
                   moves_per_s=moves_all / dt_all, model_share=model_all / dt_all,
                   env_us_per_move=1e6 * (dt_all - model_all) / (moves_all / world), games_finished=int(stats[0]))
        if args.json:
            print(json.dumps(res), flush=True)
        else:
            print("cfg5: %d GPU(s) x %d games, batch %d, df_model2-shaped net (%s, %s): %.0f moves/s end to end; "
                  "model %.1f%% of wall time, environment + glue %.1f%% (%.1f us per move); games finished %d"
                  % (world, args.num_games, args.batchsize, res["net"], res["dtype"], res["moves_per_s"],
                     100 * res["model_share"], 100 * (1 - res["model_share"]), res["env_us_per_move"],
                     res["games_finished"]), flush=True)
    GC.Stop()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
