#!/usr/bin/env python3
"""One-off differential sweep (not a test): goenv_generate with its default kernel choice against k_generate<NW> (GOENV_GEN_PIPE=0) over
odd board counts, ring depths, D4 codes, left-out outputs and ply caps; rings, final states and stats must be identical.
Usage: tools/fuzz_generate_shapes.py [cases [seed [big]]]     big = 1: board counts beyond what one launch of the pipelined kernels
holds (14,338 .. 40,000: goenv_generate in chunks, GOENV_GEN_CHUNK_MIN_PLIES=1 so that short launches are cut too)"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from elf_amd import goenv

rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
fixed = [1, 2, 3, 5, 8, 9, 63, 64, 65, 255, 1023, 1024, 1025, 1279, 1281, 1799, 1801, 2047, 2049, 3071, 3073, 5000, 8191, 8193, 14336, 14337]
big = len(sys.argv) > 3 and sys.argv[3] == "1"
if big:
    os.environ["GOENV_GEN_CHUNK_MIN_PLIES"] = "1"
    fixed = [14338, 14400, 15001, 16384, 16385, 17000, 20001, 24577, 28673, 32768, 33000, 40000]
kernels = {}
for case in range(cases):
    n = fixed[case] if case < len(fixed) else (rnd.randint(14338, 40000) if big else rnd.choice([rnd.randint(1, 300), rnd.randint(300, 3000), rnd.randint(3000, 15000)]))
    T = rnd.choice([1, 2, 3, 4]) if big else (rnd.choice([1, 2, 3, 4, 5, 7, 9, 13]) if n > 3000 else rnd.choice([1, 2, 3, 5, 8, 17, 40]))
    what = rnd.choice(["fla", "fla", "fl", "f", "la", "a", "fa"])
    aug = rnd.random() < 0.5
    cap = rnd.choice([6, 15, 40, 722])
    preset = "f" in what and rnd.random() < 0.25
    launches = rnd.choice([1, 2, 3])
    seed = rnd.randint(1, 1 << 30)
    res = []
    for pipe in (False, True):
        if pipe:
            os.environ.pop("GOENV_GEN_PIPE", None)
        else:
            os.environ["GOENV_GEN_PIPE"] = "0"
        env = goenv.GoEnv(n, seed=seed, max_ply=cap)
        env.set_constant_planes_preset(preset)
        dev = env.device
        augs = ((torch.arange(n, dtype=torch.int32, device=dev) * 3 + 1) % 8).contiguous() if aug else None
        feat = torch.full((T, n, 25, 19, 19), 0.0 if preset else -2.0, device=dev) if "f" in what else None
        legal = torch.full((T, n, 361), 5, dtype=torch.uint8, device=dev) if "l" in what else None
        acts = torch.full((T, n), -7, dtype=torch.int32, device=dev) if "a" in what else None
        got = []
        for _ in range(launches):
            env.generate(T, feat, legal, acts, aug_codes=augs)
            torch.cuda.synchronize()
            got += [x.clone() for x in (feat, legal, acts) if x is not None]
        got += [v for _, v in sorted(env.get_state().items())]
        got.append(torch.tensor(env.stats()))
        k = env.generate_kernel()
        env.close()
        res.append((got, k))
        del feat, legal, acts
        torch.cuda.empty_cache()
    for i, (x, y) in enumerate(zip(res[0][0], res[1][0])):
        if x.dtype == torch.float32:
            x, y = x.view(torch.int32), y.view(torch.int32)
        elif x.dtype == torch.uint16:
            x, y = x.view(torch.int16), y.view(torch.int16)
        assert torch.equal(x, y), ("MISMATCH", case, n, T, what, aug, cap, preset, i)
    kernels[res[1][1]] = kernels.get(res[1][1], 0) + 1
    print("case %3d: %5d boards x %2d plies x %d launches, outputs %-3s aug=%d cap=%3d preset=%d -> %s identical" % (
        case, n, T, launches, what, aug, cap, preset, res[1][1]), flush=True)
print("all %d cases identical; kernels used: %s" % (cases, kernels))
