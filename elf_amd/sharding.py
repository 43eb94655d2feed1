"""Multi-GPU sharding of the board batch.

Boards are independent, so ranks never exchange data on the step path: rank r owns the
contiguous global board ids [r*B, (r+1)*B) and keys the on-device policy RNG by the GLOBAL id,
which makes results independent of the partition.  The only collective is the end-of-episode
all-reduce of the int64 stats vector (RCCL on GPUs, gloo in the CPU tests): SUM for every slot
except MAX_LEN, which reduces with MAX.
"""
import torch
import torch.distributed as dist

ST_MAX_LEN = 9


def shard_range(total_boards, rank, world_size):
    """Contiguous shard [lo, hi) of `total_boards` for `rank`; sizes differ by at most one."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size %r/%r" % (rank, world_size))
    base, rem = divmod(total_boards, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def reduce_stats(stats, group=None):
    """All-reduce a stats vector (int64[16]) across ranks; returns a new tensor."""
    out = stats.clone()
    if not (dist.is_available() and dist.is_initialized()):
        return out
    mx = out[ST_MAX_LEN:ST_MAX_LEN + 1].clone()
    dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    out[ST_MAX_LEN] = mx[0]
    return out
