"""`sampler.DistributedEvalSampler` (scripts/inference3d_multigpu.py:34,319) does not exist in the reference.
Rank r of W gets indices r, r+W, r+2W, ... with no padding duplicates, so that concatenating the ranks' k-th items
restores the global order (relied on at inference3d_multigpu.py:371-375)."""
import torch.distributed as dist

__all__ = ['DistributedEvalSampler', 'ContiguousShardSampler']


def _rank_world(rank, num_replicas):
    if num_replicas is None:
        num_replicas = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    return rank, num_replicas


class DistributedEvalSampler:
    def __init__(self, dataset, num_replicas=None, rank=None):
        self.rank, self.num_replicas = _rank_world(rank, num_replicas)
        self.total = len(dataset)

    def __iter__(self):
        return iter(range(self.rank, self.total, self.num_replicas))

    def __len__(self):
        return len(range(self.rank, self.total, self.num_replicas))


class ContiguousShardSampler:
    """The partition empanada_amd.inference.sharded uses instead: rank r owns one contiguous block of slices."""

    def __init__(self, dataset, num_replicas=None, rank=None):
        from .inference.sharded import shard_bounds
        self.rank, self.num_replicas = _rank_world(rank, num_replicas)
        b = shard_bounds(len(dataset), self.num_replicas)
        self.lo, self.hi = int(b[self.rank]), int(b[self.rank + 1])

    def __iter__(self):
        return iter(range(self.lo, self.hi))

    def __len__(self):
        return self.hi - self.lo
