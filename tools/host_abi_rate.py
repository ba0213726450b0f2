#!/usr/bin/env python3
"""Rate of the host-pointer C entry points alone (flat buffers already built, no Python copies):
what a binding in a compiled language sees per call, PCIe transfers included."""
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
    lib, h = eng._lib, eng._h
    n, buf, distinct = 1024, 1 << 20, 16
    texts = [corpus.text(buf, 7000 + i) for i in range(distinct)]
    streams = [orc.compress(t, 6, orc.FMT_GZIP) for t in texts]
    flat = np.frombuffer(b"".join(streams[i % distinct] for i in range(n)), dtype=np.uint8)
    in_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum([len(streams[i % distinct]) for i in range(n)], out=in_off[1:])
    out_off = np.arange(n + 1, dtype=np.uint64) * buf
    out = np.empty(n * buf, dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)

    def dec():
        r = lib.rsx_batch_decompress(h, rsx.FORMAT_GZIP, flat.ctypes.data, in_off.ctypes.data, n, out.ctypes.data,
                                     out_off.ctypes.data, sizes.ctypes.data, status.ctypes.data)
        assert r == 0 and (status == 0).all()
    dec()
    t0 = time.perf_counter()
    for _ in range(3):
        dec()
    dt = (time.perf_counter() - t0) / 3
    assert out[buf:2 * buf].tobytes() == texts[1]
    print("rsx_batch_decompress (host pointers): %.2f GiB/s in+out (%d x 1 MiB gzip, %.1f ms per call)"
          % ((flat.size + out.size) / dt / 2**30, n, dt * 1e3))

    datas = [corpus.mixed(buf, 8000 + i) for i in range(distinct)]
    flat = np.frombuffer(b"".join(datas[i % distinct] for i in range(n)), dtype=np.uint8)
    in_off = np.arange(n + 1, dtype=np.uint64) * buf
    bound = rsx.compress_bound(rsx.FORMAT_ZLIB, buf)
    out_off = np.arange(n + 1, dtype=np.uint64) * bound
    out = np.empty(n * bound, dtype=np.uint8)

    def comp():
        r = lib.rsx_batch_compress(h, rsx.FORMAT_ZLIB, 6, flat.ctypes.data, in_off.ctypes.data, n, out.ctypes.data,
                                   out_off.ctypes.data, sizes.ctypes.data, status.ctypes.data)
        assert r == 0 and (status == 0).all()
    comp()
    t0 = time.perf_counter()
    for _ in range(2):
        comp()
    dt = (time.perf_counter() - t0) / 2
    print("rsx_batch_compress (host pointers): %.2f GiB/s in+out (%d x 1 MiB zlib level 6, %.1f ms per call)"
          % ((flat.size + int(sizes.sum())) / dt / 2**30, n, dt * 1e3))


if __name__ == "__main__":
    main()
