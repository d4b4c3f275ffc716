"""Iteration-oriented distributed sampler (reference data/data_sampler.py:12-65): the index space is the dataset
repeated `ratio` times so one "epoch" of the loader lasts ratio real epochs (no worker restart in between);
each rank takes every num_replicas-th entry of one epoch-seeded permutation."""
import math

import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


class DistIterSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, ratio=100):
        if num_replicas is None or rank is None:
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("DistIterSampler needs num_replicas/rank or an initialised process group")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank = dataset, num_replicas, rank
        self.epoch = 0
        self.num_samples = int(math.ceil(len(dataset) * ratio / num_replicas))
        self.total_size = self.num_samples * num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)                       # same permutation on every rank
        n = len(self.dataset)
        perm = torch.randperm(self.total_size, generator=g)
        mine = (perm[self.rank:self.total_size:self.num_replicas] % n).tolist()
        assert len(mine) == self.num_samples
        return iter(mine)

    def __len__(self):
        return self.num_samples

    def set_epoch(self, epoch):
        self.epoch = epoch
