"""Times the decoder on streams that zlib cut with Z_SYNC_FLUSH (matches reach across the joints, as in pigz's
default output): one 64 MiB stream, and 1024 copies of a 1 MiB stream (GPU box, from the repo root).
RSX_INFLATE_WHOLE=1 shows the same without cutting."""
import os
import sys
import time
import zlib

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(_ROOT, "tests")); sys.path.insert(0, _ROOT)
import numpy as np
import torch

import corpus
import rsx_pkg

rsx = rsx_pkg.load()
eng = rsx.default_engine()
dev = torch.device("cuda:0")


def sync_flushed(data, every):
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    out = b""
    for at in range(0, len(data), every):
        out += co.compress(data[at:at + every]) + co.flush(zlib.Z_SYNC_FLUSH)
    return out + co.flush()


def run(streams, sizes, label):
    n = len(streams)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(s) for s in streams], out=in_off[1:])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(sizes, out=out_off[1:])
    d_in = torch.from_numpy(np.frombuffer(b"".join(streams), dtype=np.uint8).copy()).to(dev)
    d_in_off = torch.from_numpy(in_off.view(np.int64)).to(dev)
    d_out_off = torch.from_numpy(out_off.view(np.int64)).to(dev)
    d_out = torch.empty(int(out_off[-1]), dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(n, dtype=torch.int64, device=dev)
    d_status = torch.full((n,), -7, dtype=torch.int32, device=dev)
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.batch_decompress_device(rsx.FORMAT_GZIP, d_in.data_ptr(), d_in_off.data_ptr(), n, int(in_off[-1]),
                                    int(out_off[-1]), max(sizes), d_out.data_ptr(), d_out_off.data_ptr(),
                                    d_sizes.data_ptr(), d_status.data_ptr())
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool((d_status == 0).all()) and d_sizes.cpu().tolist() == list(sizes)
    print("%s: %.2f ms, %.1f GiB/s in+out, status ok %s, parts %d, not part by part %d" %
          (label, best * 1e3, (int(in_off[-1]) + int(out_off[-1])) / best / 2**30, ok, eng.last_parts(), eng.last_whole()),
          flush=True)
    return d_out


big = corpus.text(64 << 20, 5)
d = run([sync_flushed(big, 256 << 10)], [len(big)], "one 64 MiB stream, a sync flush every 256 KiB")
assert bytes(d.cpu().numpy().tobytes()) == big
small = corpus.text(1 << 20, 6)
s = sync_flushed(small, 128 << 10)
d = run([s] * 1024, [len(small)] * 1024, "1024 x 1 MiB streams, a sync flush every 128 KiB")
assert bytes(d[:len(small)].cpu().numpy().tobytes()) == small and bytes(d[-len(small):].cpu().numpy().tobytes()) == small
