#!/usr/bin/env python3
"""cfg 3 end to end (the train_df.sh loop): offline SGF replay through this repo's `go_game.GameContext`
(`GCWrapper`, mode offline, descriptor `train`: input `s` + `offline_a`, no reply -- go/game.py:88-93)
feeding a PyTorch-ROCm policy net of the shape of the reference's df_policy (go/df_model.py:7-62:
`num_layer` x [3x3 conv -> LeakyReLU(0.1) -> BN] at `dim` channels, a final 3x3 conv to
`num_future_actions` maps, one softmax over the 361 points per future move) trained with the
MultiplePrediction loss (sum of the NLL of the next k moves).  Random init, SGD; the batch tensors stay on
the device from the feature kernel to the loss.  Reports samples/s end to end and the environment's share.

    python tools/df_train_gpu.py --list_file tests/golden/corpus/train.lst --batchsize 256 --num_games 256 --iters 40
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from elf_amd import go_loader  # noqa: E402


class PolicyNet(nn.Module):
    def __init__(self, planes=25, dim=128, layers=39, k=3):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(planes if i == 0 else dim, dim, 3, padding=1) for i in range(layers)])
        self.bns = nn.ModuleList([nn.BatchNorm2d(dim) for _ in range(layers)])
        self.final = nn.Conv2d(dim, k, 3, padding=1)

    def forward(self, s):
        for conv, bn in zip(self.convs, self.bns):
            s = bn(F.leaky_relu(conv(s), 0.1))
        return self.final(s).flatten(2)            # [B, k, 361] logits


def main():
    ap = go_loader.add_arguments()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--num_layer", type=int, default=39)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--lr", type=float, default=0.01)
    ap.set_defaults(mode="offline", gpu=0, num_games=256, batchsize=256, num_collectors=1, seed=7, T=1,
                    num_games_per_thread=300, list_file=os.path.join(ROOT, "tests", "golden", "corpus", "train.lst"))
    args = ap.parse_args()
    GC = go_loader.initialize(args)
    dev = torch.device("cuda", args.gpu)
    k = args.num_future_actions
    net = PolicyNet(dim=args.dim, layers=args.num_layer, k=k).to(dev).train()
    if args.bf16:
        net = net.to(memory_format=torch.channels_last)
    opt = torch.optim.SGD(net.parameters(), lr=args.lr, momentum=0.9)
    t_model, losses = [0.0], []
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def train(batch):
        h = batch.hist(0)
        s, target = h["s"], h["offline_a"]
        ev0.record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            logits = net(s)
        loss = sum(F.cross_entropy(logits[:, i].float(), target[:, i]) for i in range(k))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        ev1.record()
        ev1.synchronize()
        t_model[0] += ev0.elapsed_time(ev1) * 1e-3
        losses.append(float(loss.detach()))

    GC.reg_callback("train", train)
    GC.Start()
    for _ in range(5):
        GC.Run()
    torch.cuda.synchronize()
    t_model[0] = 0.0
    first = sum(losses[-3:]) / 3
    t0 = time.perf_counter()
    n = 0
    for _ in range(args.iters):
        GC.Run()
        n += len(GC.infos.s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("cfg3 train: %d games, batch %d, df_policy-shaped net (%d layers x %d ch, %s), k=%d: %.0f samples/s end to "
          "end; model fwd+bwd+step %.1f%% of wall time, environment + glue %.1f%% (%.1f us per sample); "
          "loss %.3f -> %.3f" % (args.num_games, args.batchsize, args.num_layer, args.dim,
                                 "bf16 autocast" if args.bf16 else "fp32", k, n / dt, 100 * t_model[0] / dt,
                                 100 * (1 - t_model[0] / dt), 1e6 * (dt - t_model[0]) / n, first,
                                 sum(losses[-3:]) / 3))
    GC.Stop()


if __name__ == "__main__":
    main()
