#!/usr/bin/env python3
"""Does the decoder gain from pieces of a batch that overlap on the GPU?  The bench batch (gzip, level-6 text)
decoded by one engine, and cut into 2 / 4 / 8 contiguous pieces that as many engines (a stream each, a host
thread each) decode at the same time: tools/overlap_probe.py [streams] [buf_kib].  GPU box, from the repo root."""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import bench  # noqa: E402
import rsx_pkg  # noqa: E402
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
buf = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) * 1024
rsx = rsx_pkg.load()
fmt = rsx.FORMAT_GZIP
dev = torch.device("cuda", 0)
texts, streams = bench.make_streams(16, buf, fmt, 6, 0)
steps = 8


def piece(lo, hi):
    idx = [i % 16 for i in range(lo, hi)]
    lens = np.array([len(streams[i]) for i in idx], dtype=np.uint64)
    in_off = np.zeros(len(idx) + 1, dtype=np.uint64)
    np.cumsum(lens, out=in_off[1:])
    h_in = np.concatenate([streams[i] for i in idx])
    k = hi - lo
    return dict(n=k, total_in=int(in_off[-1]), d_in=torch.from_numpy(h_in).to(dev),
                d_in_off=torch.from_numpy(in_off.view(np.int64)).to(dev),
                d_out_off=torch.from_numpy((np.arange(k + 1, dtype=np.uint64) * np.uint64(buf)).view(np.int64)).to(dev),
                d_out=torch.empty(k * buf, dtype=torch.uint8, device=dev),
                d_sizes=torch.zeros(k, dtype=torch.int64, device=dev),
                d_status=torch.full((k,), -7, dtype=torch.int32, device=dev), eng=rsx.Engine(0))


def run(p, reps):
    for _ in range(reps):
        p["eng"].batch_decompress_device(fmt, p["d_in"].data_ptr(), p["d_in_off"].data_ptr(), p["n"], p["total_in"],
                                         p["n"] * buf, buf, p["d_out"].data_ptr(), p["d_out_off"].data_ptr(),
                                         p["d_sizes"].data_ptr(), p["d_status"].data_ptr())
    p["eng"].synchronize()


for pieces in (1, 2, 4, 8):
    cuts = [n * k // pieces for k in range(pieces + 1)]
    ps = [piece(cuts[k], cuts[k + 1]) for k in range(pieces)]
    for p in ps:
        run(p, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(p, steps)) for p in ps]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = all(bool((p["d_status"] == 0).all()) for p in ps)
    total = sum(p["total_in"] + p["n"] * buf for p in ps)
    print("%d piece(s): %.2f ms per batch, %.1f GiB/s, ok=%s" % (pieces, dt * 1e3, total / dt / 2**30, ok), flush=True)
    for p in ps:
        p["eng"].close()
    del ps
    torch.cuda.empty_cache()
