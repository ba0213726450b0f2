"""Spreading one logical batch over the ranks of a torch.distributed job.

Buffers of a batch are independent (src/batch.rs:34-57, :79-100), so a batch is
cut into contiguous index ranges balanced by input bytes, every rank runs its
range on its own GPU, and only the small per-item results (sizes, status) and
three totals are exchanged: an all_gather of `u64 sizes` + `i32 status` and an
all_reduce of (bytes in, bytes out, failed items).  Under the `nccl` backend
(RCCL) those tensors live on the rank's GPU; under `gloo` on the CPU.  The
payload itself stays where it was produced unless `gather_packed` is asked to
bring it to one rank.  There is no data-path collective.
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


def _world(dist):
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def collective_device(dist):
    """Where tensors handed to collectives must live: the current GPU under nccl (RCCL
    rejects CPU tensors), the CPU under gloo."""
    import torch
    if dist is not None and dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def gather_results(sizes, status, ranges, dist, total_in=0):
    """all_gather of this rank's per-item `sizes` (int64 / uint64) and `status` (int32), given as
    torch tensors (any device) or numpy arrays, plus the all_reduce of (total_in, sum of sizes of
    the items that succeeded, failed items).

    Returns (all_sizes, all_status, totals): numpy arrays over the whole batch in item order and
    a 3-tuple of ints, identical on every rank."""
    import torch
    world, rank = _world(dist)
    dev = collective_device(dist)
    s, e = ranges[rank]

    def as_tensor(x, dtype):
        if isinstance(x, torch.Tensor):
            return x.to(device=dev, dtype=dtype)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x)).astype(
            {torch.int64: np.int64, torch.int32: np.int32}[dtype])).to(dev)

    sz = as_tensor(sizes, torch.int64)
    st = as_tensor(status, torch.int32)
    assert sz.numel() == e - s and st.numel() == e - s
    ok = st == 0
    totals = torch.stack([torch.tensor(int(total_in), dtype=torch.int64, device=dev),
                          (sz * ok).sum(), (~ok).sum().to(torch.int64)])
    if world == 1:
        t = totals.cpu().numpy()
        return sz.cpu().numpy(), st.cpu().numpy(), (int(t[0]), int(t[1]), int(t[2]))
    cap = max(b - a for a, b in ranges)
    mine_sz = torch.zeros(cap, dtype=torch.int64, device=dev)
    mine_st = torch.full((cap,), -1, dtype=torch.int32, device=dev)
    mine_sz[: e - s] = sz
    mine_st[: e - s] = st
    got_sz = [torch.empty(cap, dtype=torch.int64, device=dev) for _ in range(world)]
    got_st = [torch.empty(cap, dtype=torch.int32, device=dev) for _ in range(world)]
    dist.all_gather(got_sz, mine_sz)
    dist.all_gather(got_st, mine_st)
    dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    all_sizes = np.concatenate([got_sz[r][: b - a].cpu().numpy() for r, (a, b) in enumerate(ranges)])
    all_status = np.concatenate([got_st[r][: b - a].cpu().numpy() for r, (a, b) in enumerate(ranges)])
    t = totals.cpu().numpy()
    return all_sizes, all_status, (int(t[0]), int(t[1]), int(t[2]))


def run_sharded(items, lengths_for_balance, local_fn, dist=None):
    """Runs local_fn(sub_items, start, end) -> (results, sizes, status) on this rank's range
    (sizes / status: numpy arrays or torch tensors, one entry per item of the range) and
    gathers sizes and status from every rank.

    Returns ((start, end), my_results, all_sizes, all_status, totals) where all_* cover the
    whole batch in item order on every rank and totals = (bytes in, bytes out, failed items)."""
    world, rank = _world(dist)
    ranges = partition(lengths_for_balance, world)
    s, e = ranges[rank]
    results, sizes, status = local_fn(items[s:e], s, e)
    my_in = int(np.asarray(lengths_for_balance[s:e], dtype=np.int64).sum())
    all_sizes, all_status, totals = gather_results(sizes, status, ranges, dist, total_in=my_in)
    return (s, e), results, all_sizes, all_status, totals


def gather_packed(local, dist, root=0):
    """Brings every rank's packed output (a 1-D uint8 tensor on the collective device) to `root`:
    the lengths are all_gathered, then every other rank sends its tensor and root receives each
    one at its prefix-summed offset, all transfers posted as one batch (each peer has its own
    direct xGMI link to the root, so the transfers run side by side).  Returns the concatenation on
    root and None elsewhere."""
    import torch
    world, rank = _world(dist)
    if world == 1:
        return local
    dev = local.device
    n = torch.tensor([local.numel()], dtype=torch.int64, device=dev)
    lens = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    ops = []
    whole = None
    if rank == root:
        whole = torch.empty(int(offs[-1]), dtype=torch.uint8, device=dev)
        whole[int(offs[root]):int(offs[root + 1])] = local
        for r in range(world):
            if r != root and lens[r]:
                ops.append(dist.P2POp(dist.irecv, whole[int(offs[r]):int(offs[r + 1])], r))
    elif lens[rank]:
        ops.append(dist.P2POp(dist.isend, local, root))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return whole
