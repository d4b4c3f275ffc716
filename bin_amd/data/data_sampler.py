"""Iteration-oriented distributed sampler (reference data/data_sampler.py:12-65).

Training here is counted in iterations, not epochs, and restarting dataloader workers at every epoch boundary of a
small window list is wasted time.  So the index space is the dataset repeated `ratio` times: one pass of the loader
lasts `ratio` real epochs.  Every rank draws the SAME epoch-seeded permutation of that enlarged index space and keeps
entries rank, rank + R, rank + 2R, ... (R ranks), folded back onto real indices with a modulo — disjoint shares of
equal length whose union is `ratio` copies of the dataset (rounded up to a multiple of R)."""
import torch
import torch.distributed as dist
from torch.utils.data.sampler import Sampler


def _group(num_replicas, rank):
    if num_replicas is not None and rank is not None:
        return num_replicas, rank
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("DistIterSampler needs num_replicas/rank or an initialised process group")
    return (dist.get_world_size() if num_replicas is None else num_replicas,
            dist.get_rank() if rank is None else rank)


class DistIterSampler(Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, ratio=100):
        self.dataset = dataset
        self.num_replicas, self.rank = _group(num_replicas, rank)
        self.epoch = 0
        self.num_samples = -(-len(dataset) * ratio // self.num_replicas)          # ceil
        self.total_size = self.num_samples * self.num_replicas

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_samples

    def _share(self):
        """This rank's indices for the current epoch (a 1-D int64 tensor)."""
        gen = torch.Generator()
        gen.manual_seed(self.epoch)                                              # identical on every rank
        enlarged = torch.randperm(self.total_size, generator=gen)
        return enlarged[self.rank::self.num_replicas] % len(self.dataset)

    def __iter__(self):
        share = self._share()
        assert share.numel() == self.num_samples
        return iter(share.tolist())
