import sys, torch
sys.path.insert(0, "/root/repo")
from elf_amd import goenv
for B, T in ((1024, 64), (8192, 16)):
    for preset in (0, 1):
        env = goenv.GoEnv(B, seed=1, max_ply=722)
        env.set_constant_planes_preset(bool(preset))
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
        print("generate %5d boards x %2d plies preset=%d: %.2f us per ply, %.1f M moves/s" % (B, T, preset, ms * 1e3, B / ms / 1e3))
        env.close(); del feat, legal, moves; torch.cuda.empty_cache()
