"""k_generate_pipe (GOENV_GEN_PIPE=1: a rules wave + a feature wave per board, pipelined through an LDS snapshot) against
k_generate<2>: rings compared bit for bit from the same seed, then both timed."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from elf_amd import goenv


PACE = int(sys.argv[1]) if len(sys.argv) > 1 else 0


def make(B, pipe, aug):
    os.environ["GOENV_GEN_PIPE"] = "1" if pipe else "0"
    os.environ["GOENV_GEN_PACE_NS"] = str(PACE if pipe else 0) if PACE >= 0 else "-1"
    env = goenv.GoEnv(B, seed=7, max_ply=722)
    return env


def rings(B, T, pipe, aug, rounds):
    env = make(B, pipe, aug)
    feat = torch.full((T, B, 25, 19, 19), -3.0, device=env.device)
    legal = torch.full((T, B, 361), 7, dtype=torch.uint8, device=env.device)
    moves = torch.full((T, B), -9, dtype=torch.int32, device=env.device)
    augs = (torch.arange(B, dtype=torch.int32, device=env.device) % 8).contiguous() if aug else None
    out = []
    for _ in range(rounds):
        env.generate(T, feat, legal, moves, aug_codes=augs) if aug else env.generate(T, feat, legal, moves)
        torch.cuda.synchronize()
        out.append((feat.clone(), legal.clone(), moves.clone()))
    st = env.get_state()
    stats = env.stats()
    env.close()
    return out, st, stats


for B, T, aug, rounds in ((256, 48, False, 12), (200, 40, True, 12)):
    a, sa, ta = rings(B, T, 0, aug, rounds)
    p, sp, tp = rings(B, T, 1, aug, rounds)
    for r in range(rounds):
        for x, y, name in zip(a[r], p[r], ("planes", "legal", "moves")):
            assert torch.equal(x.view(torch.int32) if x.dtype == torch.float32 else x, y.view(torch.int32) if y.dtype == torch.float32 else y), (B, r, name)
    for k in sa:
        assert torch.equal(sa[k], sp[k]), k
    assert list(ta) == list(tp), (ta, tp)
    print("rings identical: %d boards x %d plies x %d launches, aug=%s, games finished %d" % (B, T, rounds, aug, int(ta[0])))

for B, T in ((1024, 64), (2048, 32), (4096, 16), (8192, 16)) if len(sys.argv) > 2 else ():
    for pipe in (0, 1, 0, 1):
        env = make(B, pipe, False)
        feat = torch.zeros(T, B, 25, 19, 19, device=env.device)
        legal = torch.empty(T, B, 361, dtype=torch.uint8, device=env.device)
        moves = torch.empty(T, B, dtype=torch.int32, device=env.device)
        env.ply_repeat(500, goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART)
        for _ in range(4): env.generate(T, feat, legal, moves)
        torch.cuda.synchronize()
        env.timer_begin()
        for _ in range(16): env.generate(T, feat, legal, moves)
        env.timer_end()
        ms = env.timer_ms() / (16 * T)
        print("generate %5d boards x %2d plies pipe=%d: %.2f us per ply, %.1f M moves/s" % (B, T, pipe, ms * 1e3, B / ms / 1e3), flush=True)
        env.close(); del feat, legal, moves; torch.cuda.empty_cache()
