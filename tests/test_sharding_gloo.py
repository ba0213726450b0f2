"""Multi-rank plumbing on CPU: world_size 2 over gloo.  The per-rank compute is the
oracle (no GPU here); what is under test is the partition and the gather."""
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import corpus
import oracle_lib as o
import rsx_pkg


def test_partition_properties():
    rsx = rsx_pkg.load()
    from libdeflate_rsx_amd.sharding import partition
    rs = np.random.RandomState(0)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 7, 64, 1000):
            lens = rs.randint(0, 100000, size=n)
            parts = partition(lens, world)
            assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == n
            assert all(a <= b for a, b in parts)
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            if n >= 8 * world:
                loads = [int(lens[a:b].sum()) for a, b in parts]
                assert max(loads) <= 1.5 * sum(loads) / world + 100000
    assert partition([5] * 8, 8) == [(i, i + 1) for i in range(8)]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rsx_pkg.load()
    from libdeflate_rsx_amd.sharding import run_sharded
    texts = [corpus.text(20000 + 3000 * i, i) for i in range(9)]
    streams = [o.compress(t, 6, o.FMT_GZIP) for t in texts]
    streams[4] = streams[4][:-3]  # one broken item must not disturb the others

    def local(sub, s, e):
        res, sz, st = [], [], []
        for k, c in enumerate(sub):
            r, _, out = o.decompress(c, len(texts[s + k]), o.FMT_GZIP)
            res.append(out if r == 0 else None)
            sz.append(len(out) if r == 0 else 0)
            st.append(r)
        return res, np.array(sz, dtype=np.int64), np.array(st, dtype=np.int32)

    (s, e), res, all_sizes, all_status, totals = run_sharded(streams, [len(c) for c in streams], local, dist)
    ok = all(res[k] == (texts[s + k] if s + k != 4 else None) for k in range(e - s))
    # the payload gather: every rank's packed output lands on rank 0 in item order
    import torch
    from libdeflate_rsx_amd.sharding import gather_packed
    packed = torch.from_numpy(np.frombuffer(b"".join(r for r in res if r is not None), dtype=np.uint8).copy())
    whole = gather_packed(packed, dist, root=0)
    if rank == 0:
        ok = ok and whole.numpy().tobytes() == b"".join(t for i, t in enumerate(texts) if i != 4)
    else:
        ok = ok and whole is None
    q.put((rank, s, e, ok, all_status.tolist(), all_sizes.tolist(), list(totals)))
    dist.destroy_process_group()


def test_two_ranks_cover_the_batch_and_agree():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, e0, ok0, st0, sz0, tot0), (r1, s1, e1, ok1, st1, sz1, tot1) = out
    assert (s0, e1) == (0, 9) and e0 == s1 and ok0 and ok1
    assert st0 == st1 and [x != 0 for x in st0] == [i == 4 for i in range(9)]
    want_sizes = [20000 + 3000 * i if i != 4 else 0 for i in range(9)]
    assert sz0 == sz1 == want_sizes
    assert tot0 == tot1 and tot0[1] == sum(want_sizes) and tot0[2] == 1
