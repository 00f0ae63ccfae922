"""Multi-GPU layer for batched streams: partition whole streams over ranks and
exchange per-stream compressed sizes.

Streams are independent (count and prev per stream, nothing shared), so a batch
shards at stream granularity with no data-path communication; a single stream
is never split.  The only exchange is an all-gather of the per-stream sizes
(8 bytes per stream) so that every rank can lay out the global byte directory.
Works with any torch.distributed backend ("nccl" = RCCL on GPUs, "gloo" on CPU).
"""
import numpy as np


def partition_streams(counts, world):
    """Cut stream ids 0..S-1 into `world` contiguous ranges of near-equal total
    value count.  Returns an int64 array `cut` of length world+1; rank r owns
    streams cut[r]:cut[r+1]."""
    counts = np.asarray(counts, dtype=np.int64)
    s = counts.size
    csum = np.concatenate([[0], np.cumsum(counts)])
    total = int(csum[-1])
    cut = np.zeros(world + 1, dtype=np.int64)
    for r in range(1, world):
        target = total * r // world
        # first stream boundary at or past the target
        cut[r] = min(s, max(int(np.searchsorted(csum, target, side="left")), int(cut[r - 1])))
    cut[world] = s
    return cut


def all_gather_sizes(dist, local_sizes, cut):
    """All-gather per-stream sizes (torch int64 tensor of this rank's streams).
    Shards may differ in length, so each rank pads to the longest shard; one collective into one
    flat tensor (`all_gather_into_tensor`: no list of per-rank buffers, no `cat` when the shards are
    equally long).  Returns (sizes_all, offsets_all): global per-stream sizes and their exclusive
    prefix sum (the byte directory), as torch tensors on the same device as `local_sizes`."""
    import torch
    world = dist.get_world_size()
    lens = [int(cut[r + 1] - cut[r]) for r in range(world)]
    longest = max(lens) if lens else 0
    if local_sizes.numel() == longest:
        pad = local_sizes.contiguous()
    else:
        pad = torch.zeros(longest, dtype=torch.int64, device=local_sizes.device)
        pad[: local_sizes.numel()] = local_sizes
    flat = torch.empty(world * longest, dtype=torch.int64, device=local_sizes.device)
    dist.all_gather_into_tensor(flat, pad)
    if all(n == longest for n in lens):
        sizes_all = flat
    else:
        sizes_all = torch.cat([flat[r * longest: r * longest + lens[r]] for r in range(world)])
    offsets = torch.cumsum(sizes_all, 0) - sizes_all
    return sizes_all, offsets
