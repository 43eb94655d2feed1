"""CPU: the N>1 path -- shard arithmetic and the stats all-reduce over gloo, world_size 2."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from elf_amd.sharding import reduce_stats, shard_range


def test_shard_ranges_partition_the_batch():
    for total in (8, 1024, 65536, 1000, 7):
        for world in (1, 2, 4, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
    assert shard_range(65536, 3, 8) == (3 * 8192, 4 * 8192)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = torch.arange(16, dtype=torch.int64) * (rank + 1)
    s[9] = 100 + 7 * rank  # MAX slot
    out = reduce_stats(s)
    q.put((rank, out.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_stats_allreduce_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [i * 3 for i in range(16)]
    want[9] = 107
    assert res[0] == want and res[1] == want


def test_reduce_stats_without_process_group_is_identity():
    s = torch.arange(16, dtype=torch.int64)
    assert reduce_stats(s).tolist() == s.tolist()
