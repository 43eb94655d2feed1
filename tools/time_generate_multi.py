#!/usr/bin/env python3
"""k_generate_pipe_multi (more boards than resident workgroups: several boards per workgroup, waiting in LDS) against
k_generate<NW> (GOENV_GEN_PIPE=0): rings compared bit for bit from the same seed at ragged sizes, then both timed.
Usage: tools/time_generate_multi.py [check|time]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "check"
if mode == "check":
    import torch
    from elf_amd import goenv

    def run(B, T, pipe, rounds, cap):
        if pipe:
            os.environ.pop("GOENV_GEN_PIPE", None)
        else:
            os.environ["GOENV_GEN_PIPE"] = "0"
        env = goenv.GoEnv(B, seed=11, max_ply=cap)
        augs = ((torch.arange(B, dtype=torch.int32, device=env.device) * 5) % 8).contiguous()
        feat = torch.full((T, B, 25, 19, 19), -3.0, device=env.device)
        legal = torch.full((T, B, 361), 7, dtype=torch.uint8, device=env.device)
        moves = torch.full((T, B), -9, dtype=torch.int32, device=env.device)
        cs = []
        for _ in range(rounds):
            env.generate(T, feat, legal, moves, aug_codes=augs)
            torch.cuda.synchronize()
            cs.append((int(feat.view(torch.int32).to(torch.int64).mul(torch.arange(T * B, device=env.device).view(T, B, 1, 1, 1) % 1000 + 1).sum().item()),
                       int((legal.to(torch.int64) * (torch.arange(361, device=env.device) + 1)).sum().item()), int(moves.to(torch.int64).sum().item())))
        st = env.get_state()
        out = (cs, {k: v.clone() for k, v in st.items()}, env.stats(), env.generate_kernel())
        env.close()
        return out
    for B, T, rounds, cap in ((2500, 7, 5, 30), (4100, 5, 4, 25), (8192, 4, 3, 722), (9001, 3, 3, 20)):
        a = run(B, T, False, rounds, cap)
        p = run(B, T, True, rounds, cap)
        assert p[3] == "k_generate_pipe_multi" and a[3] == "k_generate", (a[3], p[3])
        assert a[0] == p[0], (B, a[0], p[0])
        for k in a[1]:
            x, y = a[1][k], p[1][k]
            if x.dtype == torch.uint16:
                x, y = x.view(torch.int16), y.view(torch.int16)
            assert torch.equal(x, y), (B, k)
        assert list(a[2]) == list(p[2]), (a[2], p[2])
        print("rings, states and stats identical: %d boards x %d plies x %d launches (games finished %d)" % (B, T, rounds, int(a[2][0])), flush=True)
    sys.exit(0)
if mode == "--one":
    import torch
    from elf_amd import goenv
    B, T = int(sys.argv[2]), int(sys.argv[3])
    env = goenv.GoEnv(B, seed=1, max_ply=722)
    feat = torch.zeros(T, B, 25, 19, 19, device="cuda")
    legal = torch.zeros(T, B, 361, dtype=torch.uint8, device="cuda")
    moves = torch.zeros(T, B, dtype=torch.int32, device="cuda")
    env.ply_repeat(300, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
    torch.cuda.synchronize()
    out = []
    for group in range(6):
        env.timer_begin()
        for _ in range(8):
            env.generate(T, feat, legal, moves)
        env.timer_end()
        out.append(env.timer_ms() * 1e3 / (8 * T))
    pace, est = env.generate_pace()
    print("%5d boards x %3d plies %-22s us per ply by group of 8 launches: %s | pace %.2f, estimates %s" % (
        B, T, env.generate_kernel(), " ".join("%.2f" % x for x in out), pace, " ".join("%.2f" % e for e in est)), flush=True)
    sys.exit(0)
for B, T in ((1800, 32), (2048, 32), (2560, 32), (3072, 32), (4096, 32), (6144, 16), (8192, 16), (12288, 8), (16384, 8)):
    for pipe in ("0", None):
        env = dict(os.environ)
        env.pop("GOENV_GEN_PIPE", None)
        if pipe is not None:
            env["GOENV_GEN_PIPE"] = pipe
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(B), str(T)], capture_output=True, text=True, timeout=300, env=env)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-800:], flush=True)
