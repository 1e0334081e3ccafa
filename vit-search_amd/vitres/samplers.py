"""Repeated-augmentation sampler of the training loader (reference samplers.py:12-64, `--repeated-aug`, main.py:278-281).

Per epoch: a permutation of the dataset seeded by the epoch number (identical on every rank), every index repeated three
times in place, padded cyclically to a multiple of the rank count, dealt round-robin to the ranks -- so the three copies of a
sample (which the loader augments differently) land on different ranks -- and cut to floor(len // 256 * 256 / ranks) indices.
"""
import math

import torch
import torch.distributed as dist

REPEATS = 3


class RASampler(torch.utils.data.Sampler):
    def __init__(self, dataset, num_replicas=None, rank=None, shuffle=True):
        if num_replicas is None or rank is None:
            if not dist.is_available():
                raise RuntimeError("Requires distributed package to be available")
            num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
            rank = dist.get_rank() if rank is None else rank
        self.dataset, self.num_replicas, self.rank, self.shuffle = dataset, num_replicas, rank, shuffle
        self.epoch = 0
        n = len(dataset)
        self.num_samples = int(math.ceil(n * float(REPEATS) / num_replicas))
        self.total_size = self.num_samples * num_replicas
        self.num_selected_samples = int(math.floor(n // 256 * 256 / num_replicas))

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.num_selected_samples

    def __iter__(self):
        n = len(self.dataset)
        if self.shuffle:
            order = torch.randperm(n, generator=torch.Generator().manual_seed(self.epoch))
        else:
            order = torch.arange(n)
        rep = order.repeat_interleave(REPEATS)
        pad = self.total_size - rep.numel()
        if pad > 0:
            rep = torch.cat([rep, rep[:pad]])
        assert rep.numel() == self.total_size
        mine = rep[self.rank:self.total_size:self.num_replicas]
        assert mine.numel() == self.num_samples
        return iter(mine[:self.num_selected_samples].tolist())
