"""Engines over several GPUs (rsx_engine_create_multi) and the sliced host path, through the C ABI.

The test box has one GPU: an engine over devices [0, 0] runs two shards of every batch side by side on it,
each with its own stream, workspace and host thread, which is what an engine over [0..7] does on a node."""
import os
import zlib

import numpy as np
import pytest

import corpus
import oracle_lib as o
import rsx_pkg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rsx():
    return rsx_pkg.load()


def batch(rsx):
    datas = [corpus.text(300000, 1), corpus.mixed(150000, 2), b"", corpus.json_like(90000, 3), b"x",
             corpus.random_bytes(70000, 4), corpus.text(700000, 5), bytes(50000)] * 3
    return datas


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_engine_matches_the_single_engine(rsx, devices):
    single = rsx.Engine(0)
    multi = rsx.Engine(devices=devices)
    assert multi.num_devices == len(devices)
    datas = batch(rsx)
    for fmt in (rsx.FORMAT_DEFLATE, rsx.FORMAT_ZLIB, rsx.FORMAT_GZIP):
        want, wst = single.batch_compress(datas, fmt, 6)
        got, gst = multi.batch_compress(datas, fmt, 6)
        assert list(gst) == list(wst) and got == want      # byte for byte: a chunk's bytes do not depend on its GPU
        streams = [bytes(s) for s in want]
        damaged = bytearray(streams[6])
        damaged[len(damaged) // 2] ^= 0x55
        streams[6] = bytes(damaged)                          # a damaged item in the middle of a shard
        caps = [len(d) for d in datas]
        caps[0] -= 1                                         # a slot that is too small
        want, wst = single.batch_decompress(streams, caps, fmt)
        got, gst = multi.batch_decompress(streams, caps, fmt)
        assert list(gst) == list(wst) and got == want
        for i, s in enumerate(streams):
            r, _, ref = o.decompress(s, caps[i], fmt)
            assert int(gst[i]) == r and (got[i] == ref if r == 0 else got[i] is None)
        assert multi.last_fallbacks() == single.last_fallbacks()
    for kind in (rsx.CHECKSUM_CRC32, rsx.CHECKSUM_ADLER32):
        assert list(multi.batch_checksum(datas, kind)) == list(single.batch_checksum(datas, kind))
    # fewer items than devices, and none at all
    got, st = multi.batch_decompress([zlib.compress(b"abc")], [3], rsx.FORMAT_ZLIB)
    assert got == [b"abc"] and list(st) == [0]
    assert multi.batch_compress([], rsx.FORMAT_GZIP, 6)[0] == []
    multi.close()
    single.close()


def test_multi_engine_gzip_members_and_flush(rsx):
    import gzip
    multi = rsx.Engine(devices=[0, 0])
    parts = [corpus.text(200000, 11), corpus.mixed(100000, 12), b"y"]
    gz = [gzip.compress(p, 6, mtime=0) for p in parts]
    items = [gz[0] + gz[1], gz[2] * 5, gz[1], gz[0] + b"junk"]
    got, st = multi.batch_decompress_members(items, [400000] * 4)
    assert got[:3] == [parts[0] + parts[1], b"y" * 5, parts[1]] and got[3] is None and st[3] == rsx.BAD_DATA
    pieces, st = multi.batch_compress_flush(parts, 6, rsx.FLUSH_SYNC)
    last, st2 = multi.batch_compress_flush(parts, 6, rsx.FLUSH_FINISH)
    for p, a, b in zip(parts, pieces, last):
        assert zlib.decompress(a + b, -15) == p + p
    multi.close()


def test_device_entry_points_refuse_a_multi_engine(rsx):
    multi = rsx.Engine(devices=[0, 0])
    with pytest.raises(rsx.RsxError, match="single device"):
        multi.batch_checksum_device(rsx.CHECKSUM_CRC32, 0, 0, 0, 1, 0, 0, 0)
    assert multi.stream is None
    multi.close()
    assert rsx.device_count() >= 1
    with pytest.raises(rsx.RsxError):
        rsx.Engine(devices=[0, 99])


def test_host_calls_in_many_slices(rsx, monkeypatch):
    """RSX_HOST_SLICE_MB=1: a batch of a few MiB goes through the host entry points in a dozen slices (copy in,
    kernels and copy out of neighbouring slices overlap); results are those of the one-slice call and the oracle's."""
    datas = [corpus.text(400000, 21 + i) for i in range(6)] + [b"", corpus.random_bytes(300000, 30), corpus.text(2500000, 31),
                                                                corpus.mixed(600000, 32), b"z"] + [bytes([i]) * 1000 for i in range(40)]
    eng = rsx.Engine(0)
    caps = [len(d) for d in datas]
    caps[9] -= 5
    ref, ref_dec = {}, {}
    for fmt in (rsx.FORMAT_ZLIB, rsx.FORMAT_GZIP):
        ref[fmt] = eng.batch_compress(datas, fmt, 6)
        streams = list(ref[fmt][0])
        bad = bytearray(streams[2]); bad[1000] ^= 0xFF; streams[2] = bytes(bad)
        dec, dst = eng.batch_decompress(streams, caps, fmt)
        ref_dec[fmt] = (streams, dec, list(dst), eng.last_fallbacks())
        for i, s in enumerate(streams):
            r, _, want = o.decompress(s, caps[i], fmt)
            assert int(dst[i]) == r, (i, int(dst[i]), r)
            assert dec[i] == (want if r == 0 else None)
    monkeypatch.setenv("RSX_HOST_SLICE_MB", "1")
    for threads in ("0", "3"):
        monkeypatch.setenv("RSX_HOST_THREADS", threads)
        sliced = rsx.Engine(0)      # (the copy threads are started by the engine's first host call)
        for fmt in (rsx.FORMAT_ZLIB, rsx.FORMAT_GZIP):
            got, st = sliced.batch_compress(datas, fmt, 6)
            assert list(st) == list(ref[fmt][1]) and got == ref[fmt][0]
            streams, want, wst, wfb = ref_dec[fmt]
            dec, dst = sliced.batch_decompress(streams, caps, fmt)
            assert list(dst) == wst and dec == want
            assert sliced.last_fallbacks() == wfb
        crc = sliced.batch_checksum(datas, rsx.CHECKSUM_CRC32)
        assert [int(c) for c in crc] == [zlib.crc32(d) for d in datas]
        sliced.close()
    eng.close()


def test_gzip_header_crc_is_verified_on_request(rsx):
    """RSX_OPT_VERIFY_FHCRC: off (default) the FHCRC field is skipped like the reference does
    (src/decompress/mod.rs:1186-1188); on, a wrong header CRC is BadData (RFC 1952 2.3.1)."""
    text = corpus.text(120000, 41)
    body = zlib.compress(text, 6)[2:-4]
    trailer = zlib.crc32(text).to_bytes(4, "little") + (len(text) & 0xFFFFFFFF).to_bytes(4, "little")

    def member(flg, extra=b"", name=b"", good=True):
        h = bytes([0x1F, 0x8B, 8, flg, 0, 0, 0, 0, 0, 255])
        if flg & 4:
            h += len(extra).to_bytes(2, "little") + extra
        if flg & 8:
            h += name + b"\x00"
        if flg & 2:
            c = zlib.crc32(h) & 0xFFFF
            if not good:
                c ^= 0x0100
            h += c.to_bytes(2, "little")
        return h + body + trailer

    items = [member(0), member(2), member(2, good=False), member(2 | 4 | 8, b"ab" * 30, b"file.txt"),
             member(2 | 4 | 8, b"ab" * 30, b"file.txt", good=False), member(8, name=b"n")]
    caps = [len(text)] * len(items)
    eng = rsx.Engine(0)
    got, st = eng.batch_decompress(items, caps, rsx.FORMAT_GZIP)
    assert list(st) == [0] * 6 and got == [text] * 6            # the reference's behaviour
    for i, s in enumerate(items):
        assert o.decompress(s, caps[i], o.FMT_GZIP)[0] == o.SUCCESS
    eng.set_option(rsx.OPT_VERIFY_FHCRC, 1)
    got, st = eng.batch_decompress(items, caps, rsx.FORMAT_GZIP)
    assert list(st) == [0, 0, rsx.BAD_DATA, 0, rsx.BAD_DATA, 0]
    assert got == [text, text, None, text, None, text]
    # large streams take the path that cuts at joints: the header is judged there as well
    big = corpus.text(1500000, 42)
    z = o.compress(big, 6, o.FMT_GZIP)
    hdr = bytes([0x1F, 0x8B, 8, 2, 0, 0, 0, 0, 0, 255])
    ok = hdr + (zlib.crc32(hdr) & 0xFFFF).to_bytes(2, "little") + z[10:]
    ko = hdr + ((zlib.crc32(hdr) ^ 1) & 0xFFFF).to_bytes(2, "little") + z[10:]
    got, st = eng.batch_decompress([ok, ko], [len(big)] * 2, rsx.FORMAT_GZIP)
    assert list(st) == [0, rsx.BAD_DATA] and got == [big, None]
    got, st = eng.batch_decompress_members([ok + ok, ok + ko], [2 * len(big)] * 2)
    assert list(st) == [0, rsx.BAD_DATA] and got == [big + big, None]
    eng.set_option(rsx.OPT_VERIFY_FHCRC, 0)
    got, st = eng.batch_decompress([ko], [len(big)], rsx.FORMAT_GZIP)
    assert list(st) == [0] and got == [big]
    eng.close()


def test_calls_from_several_threads(rsx):
    """compress_batch / decompress_batch take &self and the reference's types are Sync (src/batch.rs:20, :74): calls
    from several threads on one engine serialise on its lock, calls on engines of their own run side by side; either
    way every caller gets its own results."""
    import threading
    shared = rsx.Engine(0)
    multi = rsx.Engine(devices=[0, 0])
    sets = [[corpus.text(200000 + 1000 * t, 100 + t), corpus.mixed(90000, 200 + t), b"", bytes([t]) * 5000] for t in range(6)]
    errors = []

    def work(t, eng):
        try:
            for fmt in (rsx.FORMAT_ZLIB, rsx.FORMAT_GZIP):
                comp, st = eng.batch_compress(sets[t], fmt, 6)
                assert list(st) == [0] * len(sets[t])
                got, st = eng.batch_decompress(comp, [len(d) for d in sets[t]], fmt)
                assert list(st) == [0] * len(sets[t]) and got == sets[t]
                crc = eng.batch_checksum(sets[t], rsx.CHECKSUM_CRC32)
                assert [int(c) for c in crc] == [zlib.crc32(d) for d in sets[t]]
        except Exception as e:   # noqa: BLE001 (reported by the main thread)
            errors.append((t, repr(e)))

    own = [rsx.Engine(0) for _ in range(2)]
    engines = [shared, shared, shared, multi, own[0], own[1]]
    threads = [threading.Thread(target=work, args=(t, engines[t])) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(120)
    assert not errors, errors
    for e in own + [shared, multi]:
        e.close()
