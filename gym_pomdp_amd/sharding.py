"""Lane sharding over the GPUs of one node.

Every lane is an independent env instance, so the batch shards embarrassingly: rank r of W owns a
contiguous range of *global* lane ids and passes its first id as `lane_offset`.  Random draws are
keyed by the global lane id, so 1, 2, 4 or 8 GPUs produce the same per-lane trajectories.  There is
no data-path collective — RCCL/xGMI are not used.  The only cross-rank traffic is the control plane
of a benchmark (a barrier and a max over ranks of a host-side timing), which runs over gloo.
"""
import os

ALIGN = 4  # the synthetic policy shares one Philox block among 4 consecutive lanes


def shard_range(total_lanes, rank, world_size, align=ALIGN):
    """(lane_offset, count) of rank's contiguous shard; boundaries are multiples of `align`, sizes
    differ by at most `align`, shards tile [0, total_lanes) exactly."""
    if not 0 <= rank < world_size:
        raise ValueError("rank %d not in [0, %d)" % (rank, world_size))
    if total_lanes < 0:
        raise ValueError("total_lanes must be >= 0")
    units = -(-total_lanes // align)                    # ceil
    lo = (units * rank // world_size) * align
    hi = (units * (rank + 1) // world_size) * align
    return min(lo, total_lanes), min(hi, total_lanes) - min(lo, total_lanes)


def env_rank():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (defaults 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def make_sharded(env_id, total_lanes, rank=None, world_size=None, **kwargs):
    """This rank's shard of a `total_lanes`-lane batched env (one process per GPU)."""
    from . import make
    r, _, w = env_rank()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    offset, count = shard_range(total_lanes, rank, world_size)
    return make(env_id, batch_size=count, lane_offset=offset, **kwargs)


class ControlPlane(object):
    """Barrier + max-over-ranks for benchmark timing.  gloo, host tensors: nothing touches xGMI."""

    def __init__(self):
        _, _, self.world_size = env_rank()
        self.dist = None
        if self.world_size > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend="gloo")
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value):
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather(self, obj):
        """[obj of rank 0, obj of rank 1, ...] on every rank (small python objects: shard ranges, labels)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world_size
        self.dist.all_gather_object(out, obj)
        return out

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()
            self.dist.destroy_process_group()
            self.dist = None
