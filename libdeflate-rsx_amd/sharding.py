"""Spreading one logical batch over the ranks of a torch.distributed job.

Buffers of a batch are independent (src/batch.rs:34-57, :79-100), so a batch is
cut into contiguous index ranges balanced by input bytes, every rank runs its
range on its own GPU, and only the small per-item results (sizes, status) are
exchanged.  There is no data-path collective.
"""
import numpy as np


def partition(lengths, world):
    """Contiguous ranges [(start, end)] * world over len(lengths) items, balanced by
    the sum of lengths; every item belongs to exactly one range."""
    n = len(lengths)
    lengths = np.asarray(lengths, dtype=np.int64)
    total = int(lengths.sum())
    if n == 0:
        return [(0, 0)] * world
    if total == 0:
        cuts = [round(n * r / world) for r in range(world + 1)]
    else:
        csum = np.concatenate([[0], np.cumsum(lengths)])
        cuts = [0]
        for r in range(1, world):
            target = total * r / world
            k = int(np.searchsorted(csum, target, side="left"))
            # choose the cut closest to the target
            if k > 0 and abs(csum[k - 1] - target) <= abs(csum[min(k, n)] - target):
                k -= 1
            cuts.append(max(cuts[-1], min(k, n)))
        cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def run_sharded(items, lengths_for_balance, local_fn, dist=None):
    """Runs local_fn(sub_items, start, end) -> (results, status ndarray) on this
    rank's range and gathers per-item status from every rank.

    Returns (my_range, my_results, all_status) where all_status covers the whole
    batch in item order on every rank.  dist is torch.distributed or None."""
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    ranges = partition(lengths_for_balance, world)
    s, e = ranges[rank]
    results, status = local_fn(items[s:e], s, e)
    status = np.asarray(status, dtype=np.int32)
    if world == 1:
        return (s, e), results, status
    import torch
    cap = max(b - a for a, b in ranges)
    mine = torch.full((cap,), -1, dtype=torch.int32)
    mine[: e - s] = torch.from_numpy(status.copy())
    gathered = [torch.empty(cap, dtype=torch.int32) for _ in range(world)]
    dist.all_gather(gathered, mine)
    all_status = np.concatenate([gathered[r][: b - a].numpy() for r, (a, b) in enumerate(ranges)])
    return (s, e), results, all_status
