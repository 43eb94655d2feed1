#!/usr/bin/env python3
"""cfg 5: the df_selfplay loop with the GPU environment.

The environment side is this repo's `go_game.GameContext` through `GCWrapper` (mode selfplay,
descriptor `actor`: input `s` (+`legal`), reply `pi`,`V` -- go/game.py:79-87); the model side is
plain PyTorch-ROCm: a policy/value ResNet of the shape of the reference's df_model2
(go/df_model2.py:28-94: 3x3 conv + BN + LeakyReLU(0.1) tower of `num_block` residual blocks at
`dim` channels, 1x1 heads, softmax policy over 361 points, tanh value), random-initialised (no
checkpoints exist here).  Moves are sampled on the device from `pi` over the legal points.
Reports moves/s end to end and the share of the step spent in the environment.

    python tools/df_selfplay_gpu.py --num_games 8192 --batchsize 2048 --iters 50
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from elf_amd import go_loader  # noqa: E402


class PolicyValueNet(nn.Module):
    def __init__(self, planes=25, dim=128, blocks=20, board=19):
        super().__init__()
        def conv(i, o, k=3, act=True):
            layers = [nn.Conv2d(i, o, k, padding=k // 2), nn.BatchNorm2d(o)]
            if act:
                layers.append(nn.LeakyReLU(0.1))
            return nn.Sequential(*layers)
        self.act = nn.LeakyReLU(0.1)
        self.stem = conv(planes, dim)
        self.tower = nn.ModuleList([nn.Sequential(conv(dim, dim), conv(dim, dim, act=False)) for _ in range(blocks)])
        d = board * board
        self.pi_head, self.pi_fc = conv(dim, 2, 1), nn.Linear(2 * d, d)
        self.v_head, self.v_fc1, self.v_fc2 = conv(dim, 1, 1), nn.Linear(d, 256), nn.Linear(256, 1)

    def forward(self, s):
        x = self.stem(s)
        for blk in self.tower:
            x = self.act(blk(x) + x)
        pi = torch.softmax(self.pi_fc(self.pi_head(x).flatten(1)), dim=1)
        v = torch.tanh(self.v_fc2(self.act(self.v_fc1(self.v_head(x).flatten(1)))))
        return pi, v


def main():
    ap = go_loader.add_arguments()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--num_block", type=int, default=20)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--bf16", action="store_true")
    ap.set_defaults(mode="selfplay", gpu=0, num_games=4096, batchsize=1024, num_collectors=0, seed=1, ply_cap=722,
                    end_on_two_passes=True, signal_restart=True)
    args = ap.parse_args()
    GC = go_loader.initialize(args)
    dev = torch.device("cuda", args.gpu)
    net = PolicyValueNet(dim=args.dim, blocks=args.num_block).to(dev).eval()
    dtype = torch.bfloat16 if args.bf16 else torch.float32
    if args.bf16:
        net = net.to(dtype).to(memory_format=torch.channels_last)
    t_model = [0.0]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def actor(batch):
        s = batch.hist(0)["s"]
        ev0.record()
        with torch.no_grad():
            pi, v = net(s.to(dtype))
        ev1.record()
        ev1.synchronize()
        t_model[0] += ev0.elapsed_time(ev1) * 1e-3
        return dict(pi=pi.float(), V=v.float().view(-1))

    GC.reg_callback("actor", actor)
    GC.Start()
    for _ in range(5):
        GC.Run()
    torch.cuda.synchronize()
    t_model[0] = 0.0
    t0 = time.perf_counter()
    moves = 0
    for _ in range(args.iters):
        GC.Run()
        moves += len(GC.infos.s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = GC.GC.Stats()
    print("cfg5: %d games, batch %d, df_model2-shaped net (%d blocks x %d ch, %s): %.0f moves/s end to end; "
          "model %.1f%% of wall time, environment + glue %.1f%% (%.1f us per move); games finished %d"
          % (args.num_games, args.batchsize, args.num_block, args.dim, "bf16" if args.bf16 else "fp32", moves / dt,
             100 * t_model[0] / dt, 100 * (1 - t_model[0] / dt), 1e6 * (dt - t_model[0]) / moves, stats[0]))
    GC.Stop()


if __name__ == "__main__":
    main()
