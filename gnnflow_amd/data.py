"""Datasets and batch samplers in front of the sampler, same classes and
arguments as gnnflow/data.py: `EdgePredictionDataset`, `RandomStartBatchSampler`,
`DistributedBatchSampler`, `default_collate_ndarray`.

Used the way scripts/offline_edge_prediction.py:190-229 does: the batch sampler yields lists
of row positions, `DataLoader(dataset, sampler=batch_sampler, collate_fn=
default_collate_ndarray)` hands each list to `dataset[list]`, which returns the batch's
(roots, ts, eid) in the layout `TemporalSampler.sample` takes.
"""
import collections.abc
import os
import re
from typing import Iterable, Iterator, List, Optional, Union

import numpy as np
import torch
import torch.distributed
from torch.utils.data import BatchSampler, Dataset, Sampler

from .utils import DstRandEdgeSampler


class EdgePredictionDataset(Dataset):
    """Positive edges of a dataframe (columns src, dst, time, eid) plus, with a negative
    sampler, one random destination per edge: item = (roots int64 [src|dst|neg],
    ts float32 [t|t|t], eid).  gnnflow/data.py:17-55."""

    def __init__(self, data, neg_sampler: Optional[DstRandEdgeSampler] = None):
        super(EdgePredictionDataset, self).__init__()
        self.data = data
        self.length = np.max(np.array(data['dst'], dtype=int))
        self.neg_sampler = neg_sampler

    def __getitem__(self, index):
        rows = self.data.iloc[index]
        src, dst, t = rows.src.values, rows.dst.values, rows.time.values
        parts = [src, dst]
        if self.neg_sampler is not None:
            parts.append(self.neg_sampler.sample(len(src)))
        roots = np.concatenate(parts).astype(np.int64)
        ts = np.concatenate([t] * len(parts)).astype(np.float32)
        return roots, ts, rows['eid'].values

    def __len__(self):
        return len(self.data)


def _chunked_batches(indices, first_size, batch_size, drop_last):
    """Lists of `batch_size` consecutive indices; when `first_size` > 0 the first list is
    that long instead (the random start of an epoch)."""
    want = first_size if first_size > 0 else batch_size
    batch = []
    for idx in indices:
        batch.append(idx)
        if len(batch) == want:
            yield batch
            batch, want = [], batch_size
    if batch and not drop_last:
        yield batch


def _local_device():
    if torch.cuda.is_available():
        return torch.device('cuda', int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device('cpu')


class RandomStartBatchSampler(BatchSampler):
    """Sequential batches whose first one is cut to a random multiple of
    `batch_size // num_chunks` each epoch, so that successive epochs see differently aligned
    batches.  gnnflow/data.py:58-118."""

    def __init__(self, sampler: Union[Sampler[int], Iterable[int]], batch_size: int,
                 drop_last: bool, num_chunks: int = 1, world_size: int = 1):
        super(RandomStartBatchSampler, self).__init__(sampler, batch_size, drop_last)
        assert 0 < num_chunks < batch_size, "num_chunks must be in (0, batch_size)"
        self.num_chunks = num_chunks
        self.chunk_size = batch_size // num_chunks
        self.reorder = self.num_chunks > 1
        self.random_size = batch_size
        # the draw is broadcast with the process group's backend when there are several ranks
        self.device = _local_device() if world_size > 1 else torch.device('cpu')
        self.world_size = world_size

    def __iter__(self) -> Iterator[List[int]]:
        self.reset()
        first = self.random_size if self.reorder else 0
        for batch in _chunked_batches(self.sampler, first, self.batch_size, self.drop_last):
            self.reorder = False
            yield batch

    def reset(self):
        self.reorder = self.num_chunks > 1
        randint = torch.randint(0, self.num_chunks, size=(1,), device=self.device)
        if self.world_size > 1:
            torch.distributed.broadcast(randint, src=0)
        self.random_size = int(randint) * self.chunk_size
        if self.random_size == 0:
            self.reorder = False


class DistributedBatchSampler(BatchSampler):
    """Rank r takes the indices with `idx % world_size == r`, batched as above; rank 0 draws
    the random start and broadcasts it.  gnnflow/data.py:121-185."""

    def __init__(self, sampler: Union[Sampler[int], Iterable[int]], batch_size: int,
                 drop_last: bool, rank: int, world_size: int, num_chunks: int = 1):
        super(DistributedBatchSampler, self).__init__(sampler, batch_size, drop_last)
        self.rank = rank
        self.world_size = world_size
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = _local_device()
        assert 0 < num_chunks < batch_size, "num_chunks must be in (0, batch_size)"
        self.num_chunks = num_chunks
        self.chunk_size = batch_size // num_chunks
        self.reorder = False
        self.random_size = batch_size

    def __iter__(self) -> Iterator[List[int]]:
        self.reset()
        mine = (idx for idx in self.sampler if idx % self.world_size == self.rank)
        first = self.random_size if self.reorder else 0
        for batch in _chunked_batches(mine, first, self.batch_size, self.drop_last):
            self.reorder = False
            yield batch

    def reset(self):
        self.reorder = self.num_chunks > 1
        if not self.reorder:
            return
        if self.rank == 0:
            randint = torch.randint(0, self.num_chunks, size=(1,), device=self.device)
        else:
            randint = torch.zeros(1, dtype=torch.int64, device=self.device)
        torch.distributed.broadcast(randint, src=0)
        self.random_size = int(randint.item() * self.chunk_size)
        if self.random_size == 0:
            self.reorder = False


_STRING_OR_OBJECT = re.compile(r'[SaUO]')


def default_collate_ndarray(batch):
    """torch's default_collate with numpy outputs (gnnflow/data.py:193-293): arrays are
    stacked and flattened column-major (so a batch of one array is that array), numbers
    become arrays, strings stay lists, mappings / namedtuples / sequences recurse."""
    elem = batch[0]
    kind = type(elem)
    if isinstance(elem, np.ndarray):
        if kind.__name__ in ('ndarray', 'memmap') and \
                _STRING_OR_OBJECT.search(elem.dtype.str) is not None:
            raise TypeError("default_collate_ndarray: batch must contain tensors, numpy arrays, "
                            "numbers, dicts or lists; found {}".format(elem.dtype))
        return np.stack(batch, 0).flatten('F')
    if kind.__module__ == 'numpy' and kind.__name__ not in ('str_', 'string_'):
        if getattr(elem, 'shape', None) == ():
            return np.array(batch)
    if isinstance(elem, float):
        return np.array(batch, dtype=np.float64)
    if isinstance(elem, int):
        return np.array(batch)
    if isinstance(elem, (str, bytes)):
        return batch
    if isinstance(elem, collections.abc.Mapping):
        merged = {key: default_collate_ndarray([d[key] for d in batch]) for key in elem}
        try:
            return kind(merged)
        except TypeError:
            return merged
    if isinstance(elem, tuple) and hasattr(elem, '_fields'):
        return kind(*(default_collate_ndarray(samples) for samples in zip(*batch)))
    if isinstance(elem, collections.abc.Sequence):
        size = len(elem)
        if any(len(e) != size for e in batch):
            raise RuntimeError('each element in list of batch should be of equal size')
        columns = [default_collate_ndarray(samples) for samples in zip(*batch)]
        if isinstance(elem, tuple):
            return columns
        try:
            return kind(columns)
        except TypeError:
            return columns
    raise TypeError("default_collate_ndarray: batch must contain tensors, numpy arrays, numbers, "
                    "dicts or lists; found {}".format(kind))
