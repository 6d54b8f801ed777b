"""Sharding primitive of the hot path (API mirror of reference misc/torchutils.py:66-68).  The
optimisers of that file are training-side and out of scope."""
import numpy as np
from torch.utils.data import Subset


def split_dataset(dataset, n_splits):
    """Strided shards: shard i holds items i, i+n, i+2n, ...  (one shard per GPU; no overlap, no
    communication between shards)."""
    return [Subset(dataset, np.arange(i, len(dataset), n_splits)) for i in range(n_splits)]


def shard_indices(n_items, rank, world_size):
    """Indices of `rank`'s strided shard — what split_dataset(...)[rank] iterates over."""
    return np.arange(rank, n_items, world_size)
