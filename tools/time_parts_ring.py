#!/usr/bin/env python3
"""Launch-per-ply steps with the batch cut into parts on streams of their own (as tools/time_parts.py), but with the
outputs going to slot t % R of a ring of R buffers: R = 1 rewrites the same buffer every ply (at most 4,096 boards of it
fit the 256-MB Infinity Cache), R = 8 does not.  Launches are enqueued from Python (prebuilt ctypes arguments), one stream per part.
Usage: tools/time_parts_ring.py [boards ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from elf_amd import goenv  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096]
    flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
    for B in sizes:
        for R in (1, 8):
            for parts in (1, 2, 3):
                env = goenv.GoEnv(B, seed=1, max_ply=722)
                feat = torch.empty(R, B, 25, 19, 19, device=env.device)
                legal = torch.empty(R, B, 361, dtype=torch.uint8, device=env.device)
                streams = [torch.cuda.Stream(device=env.device) for _ in range(parts)]
                cut = [B * p // parts for p in range(parts + 1)]

                # ctypes argument lists built once: the loop below is the C call and nothing else
                import ctypes as C
                calls = []
                for r in range(R):
                    for p in range(parts):
                        f, l = feat[r, cut[p]:cut[p + 1]], legal[r, cut[p]:cut[p + 1]]
                        calls.append((env._h, None, cut[p], cut[p + 1] - cut[p], flags, None, None, C.c_void_p(f.data_ptr()),
                                      C.c_void_p(l.data_ptr()), None, None, None, C.c_void_p(streams[p].cuda_stream)))
                fn = env.lib.goenv_ply_sel

                def run(K, t0=0):
                    for t in range(t0, t0 + K):
                        base = (t % R) * parts
                        for p in range(parts):
                            if fn(*calls[base + p]):
                                raise RuntimeError(env.lib.goenv_last_error().decode())

                env.ply_repeat(200, flags)
                torch.cuda.synchronize()
                run(100)
                torch.cuda.synchronize()
                K = 1000
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for s in streams:
                    s.wait_event(e0)
                run(K, 100)
                for s in streams:
                    torch.cuda.current_stream().wait_stream(s)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / K
                print("%6d boards, ring of %d, %d parts: %8.2f us/ply  %7.1f Mmoves/s  planes+mask %6.0f GB/s" % (
                    B, R, parts, us, B / us, B * 36461 / us / 1e3), flush=True)
                env.close()
                del feat, legal


if __name__ == "__main__":
    main()
