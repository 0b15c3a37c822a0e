"""Multi-GPU layout of the path: clips are independent, so they shard across
ranks with no data-path collective (reference parallelism: multiprocessing.Pool
over files, track/trackextractor.py:80-85).  The only exchange is one
all-gather of fixed-width per-clip result records so that every rank (or
rank 0) can write the metadata."""

import numpy as np


def partition_clips(frame_counts, world_size):
    """Greedy longest-processing-time partition of clips by frame count.
    -> list (per rank) of clip indices, each ascending; deterministic."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(frame_counts[i])
    return [sorted(s) for s in shards]


def rank_world():
    """(rank, world_size, local_rank) of this process under torchrun / torch.distributed.run, (0, 1, 0) otherwise."""
    import os

    try:
        return (int(os.environ.get("RANK", 0)), max(1, int(os.environ.get("WORLD_SIZE", 1))),
                int(os.environ.get("LOCAL_RANK", 0)))
    except ValueError:
        return 0, 1, 0


def shard_files(paths, rank, world_size):
    """This rank's share of a list of recordings: longest-processing-time partition by file size (a CPTV file's size
    tracks its frame count), identical on every rank; order of `paths` is kept inside a shard."""
    import os

    if world_size <= 1:
        return list(paths)
    sizes = [os.path.getsize(p) for p in paths]
    return [paths[i] for i in partition_clips(sizes, world_size)[rank]]


def gather_records(records, dist=None, device=None):
    """records: int32 tensor [n_local, width] (clip id in column 0).  Pads every
    rank to the global maximum with -1 rows, all_gathers once, returns the
    concatenation without padding, sorted by clip id (same on every rank)."""
    import torch

    if dist is None or not dist.is_initialized():
        out = records
    else:
        world = dist.get_world_size()
        n = torch.tensor([records.shape[0]], dtype=torch.int64, device=records.device)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n)
        cap = int(max(int(c.item()) for c in counts))
        width = records.shape[1]
        padded = torch.full((cap, width), -1, dtype=records.dtype, device=records.device)
        padded[: records.shape[0]] = records
        gathered = torch.empty((world * cap, width), dtype=records.dtype, device=records.device)
        dist.all_gather_into_tensor(gathered, padded)
        out = gathered[gathered[:, 0] >= 0]
    order = torch.argsort(out[:, 0], stable=True)
    return out[order]
