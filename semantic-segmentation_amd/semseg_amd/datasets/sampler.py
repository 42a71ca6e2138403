"""Rank sharding of an epoch's sample list: drop-in for the reference's
datasets/sampler.py:57-105 (`DistributedSampler`).  Integer work, bit-exact:
the same permutation (torch.randperm seeded with the epoch), the same padding
to a multiple of the world size and the same strided / consecutive split."""
import math

import torch
import torch.distributed as dist
from torch.utils.data import Sampler


def shard_indices(n, epoch, rank, world, pad=False, consecutive_sample=False, permutation=False, per_rank=None):
    """Indices of `rank`'s share of a dataset of `n` samples in `epoch`.  per_rank: the share length if the
    caller fixed it (DistributedSampler.set_num_samples), otherwise ceil (pad) / floor of n / world."""
    if per_rank is None:
        per_rank = int(math.ceil(n / world)) if pad else n // world
    total = per_rank * world
    if permutation:
        g = torch.Generator()
        g.manual_seed(epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    if total > n:                               # wrap around to an even split
        order = order + order[:total - n]
    if consecutive_sample:
        mine = order[per_rank * rank:per_rank * (rank + 1)]
    else:
        mine = order[rank:total:world]
    assert len(mine) == per_rank
    return mine


class DistributedSampler(Sampler):
    def __init__(self, dataset, pad=False, consecutive_sample=False, permutation=False, num_replicas=None,
                 rank=None):
        self.dataset = dataset
        self.num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
        self.rank = dist.get_rank() if rank is None else rank
        self.pad = pad
        self.consecutive_sample = consecutive_sample
        self.permutation = permutation
        self.epoch = 0
        n = len(dataset)
        self.num_samples = int(math.ceil(n / self.num_replicas)) if pad else n // self.num_replicas
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        # the share length is self.num_samples, as in the reference's __iter__ (datasets/sampler.py:78-100): after
        # set_num_samples() that is the ceiling also for a sampler built with pad=False
        return iter(shard_indices(len(self.dataset), self.epoch, self.rank, self.num_replicas, self.pad,
                                  self.consecutive_sample, self.permutation, per_rank=self.num_samples))

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch

    def set_num_samples(self):
        """datasets/sampler.py:108-110: after build_epoch() changed the dataset -- always the ceiling,
        whatever `pad` was at construction (the reference's own behaviour)."""
        self.num_samples = int(math.ceil(len(self.dataset) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas
