#!/usr/bin/env python3
"""Per-ply time of the fused step with the batch cut into 1..8 parts on streams of their own
(goenv_ply_repeat_parts), plus a check that the boards end up identical.
Usage: tools/time_parts.py [boards ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from elf_amd import goenv  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1024]
    for B in sizes:
        ref_state = None
        for parts in (1, 2, 4, 8):
            env = goenv.GoEnv(B, seed=1, max_ply=722)
            feat = env.new(torch.float32, 25, 19, 19)
            legal = env.new(torch.uint8, 361)
            flags = goenv.F_RANDOM_POLICY | goenv.F_AUTO_RESTART
            env.ply_repeat_parts(200, parts, flags, feat_out=feat, legal_out=legal)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K = 2000
            e0.record()
            env.ply_repeat_parts(K, parts, flags, feat_out=feat, legal_out=legal)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / K
            st = env.get_state()
            key = (st["color"].cpu(), st["scalars"].cpu(), feat.cpu())
            same = "-"
            if ref_state is None:
                ref_state = key
            else:
                same = all(torch.equal(a, b) for a, b in zip(ref_state, key))
            print("%6d boards %d parts: %8.2f us/ply  %7.1f Mmoves/s  %6.0f GB/s  same=%s" % (
                B, parts, us, B / us, B * 39725 / us / 1e3, same), flush=True)
            env.close()


if __name__ == "__main__":
    main()
