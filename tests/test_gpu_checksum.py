"""GPU CRC-32 / Adler-32 kernels vs the oracle, the reference's KATs and zlib."""
import zlib

import numpy as np
import pytest

import corpus
import oracle_lib as o
import rsx_pkg
from golden.expand import expand

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rsx():
    return rsx_pkg.load()


@pytest.fixture(scope="module")
def eng(rsx):
    return rsx.default_engine()


def test_kat(rsx, eng, kat):
    items = [expand(v["data"]) for v in kat["adler32"]]
    got = eng.batch_checksum(items, rsx.CHECKSUM_ADLER32)
    assert [int(x) for x in got] == [v["value"] for v in kat["adler32"]]
    items = [expand(v["data"]) for v in kat["crc32"]]
    got = eng.batch_checksum(items, rsx.CHECKSUM_CRC32)
    assert [int(x) for x in got] == [v["value"] for v in kat["crc32"]]
    assert rsx.crc32(0, b"Hello, World!") == 0xEC4AC3D0
    assert rsx.adler32(1, b"Hello, World!") == 530449514


def test_ragged_sizes_and_alignment(rsx, eng):
    # every size class around the 16-byte unit and the 32 KiB segment, at every
    # start alignment (items are packed back to back, so starts are unaligned)
    rs = np.random.RandomState(3)
    sizes = list(range(0, 70)) + [255, 256, 4095, 4096, 4097, 5552, 5553, 32767, 32768, 32769,
                                  65535, 65536, 65537, 100000, 1000003]
    items = [rs.randint(0, 256, size=s, dtype=np.uint8).tobytes() for s in sizes]
    crc = eng.batch_checksum(items, rsx.CHECKSUM_CRC32)
    adl = eng.batch_checksum(items, rsx.CHECKSUM_ADLER32)
    for i, it in enumerate(items):
        assert int(crc[i]) == o.crc32(it) == zlib.crc32(it), sizes[i]
        assert int(adl[i]) == o.adler32(it) == zlib.adler32(it), sizes[i]


def test_uniform_batch_and_overflow_patterns(rsx, eng):
    items = [b"\xff" * 1000000, b"\x00" * 300000, corpus.text(1 << 20, 1), b"\xff" * 100000]
    crc = eng.batch_checksum(items, rsx.CHECKSUM_CRC32)
    adl = eng.batch_checksum(items, rsx.CHECKSUM_ADLER32)
    for i, it in enumerate(items):
        assert int(crc[i]) == o.crc32(it)
        assert int(adl[i]) == o.adler32(it)
    items = [corpus.random_bytes(65536, s) for s in range(64)]
    crc = eng.batch_checksum(items, rsx.CHECKSUM_CRC32)
    assert [int(x) for x in crc] == [o.crc32(it) for it in items]


def test_chaining(rsx):
    d = corpus.random_bytes(200000, 9)
    assert rsx.crc32(rsx.crc32(0, d[:77777]), d[77777:]) == o.crc32(d)
    assert rsx.adler32(rsx.adler32(1, d[:77777]), d[77777:]) == o.adler32(d)


def test_device_entry_point(rsx, eng):
    # rsx_batch_checksum_device on device buffers: the caller's `max_item` only sizes the launch; values
    # are right whether it is exact, generous or understated
    import torch
    dev = torch.device("cuda", 0)
    items = [corpus.text(100000, 1), b"", corpus.random_bytes(32768, 2), corpus.mixed(70001, 3), b"x"]
    n = len(items)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[0] = 77   # (the batch does not start at the buffer's first byte)
    off[1:] = 77 + np.cumsum([len(it) for it in items])
    h = np.concatenate([np.zeros(77, dtype=np.uint8), np.frombuffer(b"".join(items), dtype=np.uint8)])
    d_in = torch.from_numpy(h.copy()).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_val = torch.zeros(n, dtype=torch.int32, device=dev)
    e2 = rsx.Engine(0, torch.cuda.current_stream(dev).cuda_stream)
    for kind, ref, init in ((rsx.CHECKSUM_CRC32, zlib.crc32, 0), (rsx.CHECKSUM_ADLER32, zlib.adler32, 1)):
        for max_item in (100000, 1 << 20, 4096):
            d_val.zero_()
            e2.batch_checksum_device(kind, init, d_in.data_ptr(), d_off.data_ptr(), n, int(off[-1] - off[0]), max_item,
                                     d_val.data_ptr())
            torch.cuda.synchronize(dev)
            got = d_val.cpu().numpy().view(np.uint32)
            assert [int(x) for x in got] == [ref(it) for it in items], (kind, max_item)
