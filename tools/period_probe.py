"""Times the decoder on data whose matches all lie at one distance (a random block of P bytes repeated): the
worst case for sources that have left the expand kernel's LDS ring (GPU box, from the repo root).
usage: python tools/period_probe.py [P ...]"""
import os
import sys
import time
import zlib

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import numpy as np
import torch

import rsx_pkg

rsx = rsx_pkg.load()
eng = rsx.default_engine()
dev = torch.device("cuda:0")
periods = [int(a) for a in sys.argv[1:]] or [3000, 6000, 12000, 20000, 30000]
n, size = 2048, 1 << 20
for period in periods:
    rs = np.random.RandomState(period)
    distinct = []
    for k in range(8):
        block = rs.randint(0, 256, size=period, dtype=np.uint8).tobytes()
        data = (block * (size // period + 1))[:size]
        distinct.append((data, zlib.compress(data, 6)))
    streams = [distinct[i % 8][1] for i in range(n)]
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(size)
    d_in = torch.from_numpy(np.frombuffer(b"".join(streams), dtype=np.uint8).copy()).to(dev)
    d_in_off = torch.from_numpy(in_off.view(np.int64)).to(dev)
    d_out_off = torch.from_numpy(out_off.view(np.int64)).to(dev)
    d_out = torch.empty(n * size, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.batch_decompress_device(rsx.FORMAT_ZLIB, d_in.data_ptr(), d_in_off.data_ptr(), n, int(in_off[-1]), n * size,
                                    size, d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(), d_status.data_ptr())
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool((d_status == 0).all()) and bytes(d_out[:size].cpu().numpy().tobytes()) == distinct[0][0]
    print("period %5d: %.2f ms, %.1f GiB/s out, ok %s" % (period, best * 1e3, n * size / best / 2**30, ok), flush=True)
