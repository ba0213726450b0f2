#!/usr/bin/env python3
"""Decodes a full-occupancy gzip batch into a device buffer that has slack on both sides of the output
(an out-of-range access by a kernel under development then reads or writes the slack instead of faulting
the GPU) and checks every stream against its text and the slack against its fill pattern."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import corpus  # noqa: E402
import oracle_lib as o  # noqa: E402
import rsx_pkg  # noqa: E402

rsx = rsx_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
buf = 1 << 20
pad = 64 << 20
dev = torch.device("cuda", 0)
distinct = [corpus.text(buf, 900 + i) for i in range(16)]
streams = [np.frombuffer(o.compress(d, 6, 2), dtype=np.uint8) for d in distinct]
lens = np.array([len(streams[i % 16]) for i in range(n)], dtype=np.uint64)
in_off = np.zeros(n + 1, dtype=np.uint64)
np.cumsum(lens, out=in_off[1:])
h_in = np.concatenate([streams[i % 16] for i in range(n)])
out_off = np.arange(n + 1, dtype=np.uint64) * np.uint64(buf)
d_in = torch.from_numpy(h_in).to(dev)
d_in_off = torch.from_numpy(in_off.view(np.int64)).to(dev)
d_out_off = torch.from_numpy(out_off.view(np.int64)).to(dev)
whole = torch.full((n * buf + 2 * pad,), 0x5A, dtype=torch.uint8, device=dev)
d_out = whole[pad:pad + n * buf]
d_sizes = torch.zeros(n, dtype=torch.int64, device=dev)
d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
eng = rsx.Engine(0, torch.cuda.current_stream(dev).cuda_stream)
for it in range(3):
    eng.batch_decompress_device(rsx.FORMAT_GZIP, d_in.data_ptr(), d_in_off.data_ptr(), n, int(in_off[-1]),
                                n * buf, buf, d_out.data_ptr(), d_out_off.data_ptr(), d_sizes.data_ptr(),
                                d_status.data_ptr())
    torch.cuda.synchronize(dev)
    st = d_status.cpu().numpy()
    sz = d_sizes.cpu().numpy()
    bad = np.nonzero((st != 0) | (sz != buf))[0]
    print("pass", it, "bad streams", len(bad), bad[:8], st[bad[:8]], flush=True)
    assert bool((whole[:pad] == 0x5A).all()) and bool((whole[pad + n * buf:] == 0x5A).all()), "slack was written"
    assert len(bad) == 0
want = [torch.from_numpy(np.frombuffer(d, dtype=np.uint8).copy()).to(dev) for d in distinct]
rows = d_out.view(n, buf)
for i in range(n):
    assert bool((rows[i] == want[i % 16]).all()), i
print("padded check ok: %d streams" % n)
