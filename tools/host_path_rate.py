#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry points (rsx_batch_decompress / rsx_batch_compress
with host buffers): staging, kernels and the copies back, per call.  Not the headline
number (bench.py times device-resident batches); printed for the design notes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import corpus          # noqa: E402
import oracle_lib as orc  # noqa: E402
import rsx_pkg         # noqa: E402


def main():
    rsx = rsx_pkg.load()
    eng = rsx.default_engine()
    n, buf, distinct = 512, 1 << 20, 16
    texts = [corpus.text(buf, 7000 + i) for i in range(distinct)]
    streams = [orc.compress(t, 6, orc.FMT_GZIP) for t in texts]
    items = [streams[i % distinct] for i in range(n)]
    caps = [buf] * n
    eng.batch_decompress(items, caps, rsx.FORMAT_GZIP)
    t0 = time.perf_counter()
    out, st = eng.batch_decompress(items, caps, rsx.FORMAT_GZIP)
    dt = time.perf_counter() - t0
    assert (np.asarray(st) == 0).all() and out[1] == texts[1]
    nbytes = sum(len(s) for s in items) + n * buf
    print("host-path gzip decompress: %.2f GiB/s in+out (%d x 1 MiB, %.1f ms)" % (nbytes / dt / 2**30, n, dt * 1e3))
    datas = [corpus.mixed(buf, 8000 + i) for i in range(distinct)]
    items = [datas[i % distinct] for i in range(n)]
    eng.batch_compress(items, rsx.FORMAT_ZLIB, 6)
    t0 = time.perf_counter()
    comp, st = eng.batch_compress(items, rsx.FORMAT_ZLIB, 6)
    dt = time.perf_counter() - t0
    assert (np.asarray(st) == 0).all()
    nbytes = n * buf + sum(len(c) for c in comp)
    print("host-path zlib level-6 compress: %.2f GiB/s in+out (%d x 1 MiB, %.1f ms)" % (nbytes / dt / 2**30, n, dt * 1e3))


if __name__ == "__main__":
    main()
