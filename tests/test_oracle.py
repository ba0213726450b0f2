"""CPU restatement (oracle) pinned against the reference's own known-answer
vectors (tests/golden/reference_kat.json) and cross-checked against zlib."""
import os
import random
import zlib

import numpy as np
import pytest

import oracle_lib as o
from golden.expand import expand

WBITS = {o.FMT_DEFLATE: -15, o.FMT_ZLIB: 15, o.FMT_GZIP: 31}


def test_adler32_kat(kat):
    for v in kat["adler32"]:
        assert o.adler32(expand(v["data"])) == v["value"]


def test_crc32_kat(kat):
    for v in kat["crc32"]:
        assert o.crc32(expand(v["data"])) == v["value"]


def test_checksums_vs_zlib_tail_sizes():
    # tests/unit_tests.rs:352-368 sweeps tail sizes; tests/adler32_overflow.rs uses 0xFF runs
    rnd = random.Random(7)
    for n in list(range(0, 70)) + [255, 256, 4095, 4096, 4097, 5552, 5553, 100000]:
        d = bytes(rnd.getrandbits(8) for _ in range(n))
        assert o.crc32(d) == zlib.crc32(d)
        assert o.adler32(d) == zlib.adler32(d)
    for n in (100000, 1000000):
        d = b"\xff" * n
        assert o.adler32(d) == zlib.adler32(d)
        assert o.crc32(d) == zlib.crc32(d)
    # chaining
    d = os.urandom(10000)
    assert o.crc32(d[5000:], o.crc32(d[:5000])) == zlib.crc32(d)
    assert o.adler32(d[5000:], o.adler32(d[:5000])) == zlib.adler32(d)


def test_inflate_kat(kat):
    for v in kat["inflate"]:
        want = expand(v["output"])
        r, ic, out = o.decompress(bytes.fromhex(v["stream_hex"]), len(want))
        assert r == o.SUCCESS and out == want
    for v in kat["inflate_failures"]:
        r, _, _ = o.decompress(bytes.fromhex(v["stream_hex"]), v["max_out"])
        assert r != o.SUCCESS


def test_matchfinder_kat(kat):
    for v in kat["matchfinder"]:
        assert o.hc_find_match(expand(v["data"]), v["pos"], v["max_depth"], v["nice_len"]) == \
            (v["len"], v["offset"])


def _corpus():
    rnd = random.Random(1)
    words = [bytes(rnd.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(rnd.randint(2, 10)))
             for _ in range(1500)]
    yield b"a"
    yield b"abc" * 1000
    yield bytes(range(256)) * 100
    yield b" ".join(rnd.choice(words) for _ in range(60000))
    yield bytes((i * 7) % 251 for i in range(300000))  # > 256 KiB: chunked + sync flush
    yield bytes(10000)


@pytest.mark.parametrize("level", [1, 2, 4, 6, 8, 9])
def test_roundtrip_all_formats(level):
    for t in _corpus():
        for fmt in (o.FMT_DEFLATE, o.FMT_ZLIB, o.FMT_GZIP):
            c = o.compress(t, level, fmt)
            assert c is not None
            assert zlib.decompress(c, WBITS[fmt]) == t
            r, ic, out = o.decompress(c, len(t), fmt)
            assert (r, ic, out) == (o.SUCCESS, len(c), t)


def test_level0_stored():
    t = os.urandom(70000)
    c = o.compress(t, 0, o.FMT_DEFLATE)
    assert zlib.decompress(c, -15) == t and len(c) == len(t) + 10


def test_inflate_zlib_streams():
    for t in _corpus():
        for lvl in (1, 6, 9):
            for fmt in (o.FMT_DEFLATE, o.FMT_ZLIB, o.FMT_GZIP):
                co = zlib.compressobj(lvl, zlib.DEFLATED, WBITS[fmt])
                z = co.compress(t) + co.flush()
                assert o.decompress(z, len(t), fmt) == (o.SUCCESS, len(z), t)
    # stored blocks and fixed-Huffman blocks
    t = os.urandom(3000)
    z = zlib.compress(t, 0)
    assert o.decompress(z, len(t), o.FMT_ZLIB)[2] == t
    co = zlib.compressobj(6, zlib.DEFLATED, -15, 9, zlib.Z_FIXED)
    z = co.compress(b"fixed huffman block " * 50) + co.flush()
    assert o.decompress(z, 1000, o.FMT_DEFLATE)[2] == b"fixed huffman block " * 50


def test_result_codes():
    t = b"Hello world!" * 20
    c = o.compress(t, 6, o.FMT_DEFLATE)
    assert o.decompress(c, len(t) - 1)[0] == o.INSUFFICIENT_SPACE
    assert o.decompress(c[:len(c) // 2], len(t))[0] == o.SHORT_INPUT
    assert o.decompress(b"", 10)[0] == o.SHORT_INPUT
    assert o.decompress(b"\x07", 10)[0] == o.BAD_DATA          # reserved block type 3
    assert o.decompress(b"\x01\x05\x00\x00\x00", 10)[0] == o.BAD_DATA  # LEN != ~NLEN
    assert o.decompress(b"\x01\x05\x00\xfa\xffabc", 10)[0] == o.SHORT_INPUT
    assert o.decompress(b"\x01\x05\x00\xfa\xffabcde", 4)[0] == o.INSUFFICIENT_SPACE
    # match that reaches before the start of the output: fixed-Huffman block whose
    # first symbol is a length-3 match at offset 1 (bits: 1, 01, 0000001, 00000)
    assert o.decompress(bytes([0x03, 0x02, 0x00, 0x00]), 100)[0] == o.BAD_DATA
    g = o.compress(t, 6, o.FMT_GZIP)
    corrupt = bytearray(g)
    corrupt[-6] ^= 1
    assert o.decompress(bytes(corrupt), len(t), o.FMT_GZIP)[0] == o.BAD_DATA
    corrupt = bytearray(g)
    corrupt[-1] ^= 1
    assert o.decompress(bytes(corrupt), len(t), o.FMT_GZIP)[0] == o.BAD_DATA
    zl = o.compress(t, 6, o.FMT_ZLIB)
    corrupt = bytearray(zl)
    corrupt[-1] ^= 1
    assert o.decompress(bytes(corrupt), len(t), o.FMT_ZLIB)[0] == o.BAD_DATA
    assert o.decompress(b"\x78\x9d" + zl[2:], len(t), o.FMT_ZLIB)[0] == o.BAD_DATA
    assert o.decompress(zl[:5], len(t), o.FMT_ZLIB)[0] == o.SHORT_INPUT


def test_dynamic_headers_built_by_hand():
    """The oracle on block headers assembled bit by bit (tests/handmade.py): what zlib decodes it decodes
    to the same bytes; a run across the end of the code lengths is clamped (src/decompress/mod.rs:462-467;
    zlib rejects it); a repeat with nothing in front of it is BadData, or ShortInput when the input ends inside
    its codeword (:436-456 checks the bits first); a header cut anywhere is ShortInput."""
    import handmade
    streams, caps, text = handmade.header_cases()
    for k in (0, 1, 2):
        r, _, out = o.decompress(streams[k], caps[k], 0)
        assert r == o.SUCCESS and out == zlib.decompress(streams[k], -15)
    assert o.decompress(streams[0], caps[0], 0)[2] == text
    r, _, out = o.decompress(streams[3], caps[3], 0)
    assert r == o.SUCCESS and out == bytes(range(200))
    with pytest.raises(zlib.error):
        zlib.decompress(streams[3], -15)
    assert o.decompress(streams[4], caps[4], 0)[0] == o.BAD_DATA
    assert o.decompress(streams[5], caps[5], 0)[0] == o.SHORT_INPUT
    for s_, c_ in zip(streams[6:], caps[6:]):
        assert o.decompress(s_, c_, 0)[0] == o.SHORT_INPUT


def test_headers():
    # src/compress/mod.rs:2257-2269 (zlib level hint, FCHECK) and :2309-2327 (gzip XFL/OS)
    assert o.compress(b"x", 6, o.FMT_ZLIB)[:2] == b"\x78\x9c"
    assert o.compress(b"x", 1, o.FMT_ZLIB)[:2] == b"\x78\x01"
    assert o.compress(b"x", 9, o.FMT_ZLIB)[:2] == b"\x78\xda"
    assert o.compress(b"x", 6, o.FMT_GZIP)[:10] == bytes([0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 255])
    assert o.compress(b"x", 1, o.FMT_GZIP)[8] == 4 and o.compress(b"x", 9, o.FMT_GZIP)[8] == 2


def test_sync_flush_chunks():
    # inputs above 256 KiB are cut into independently compressed chunks joined by
    # sync-flush markers (src/compress/mod.rs:699-772, :662-681)
    t = bytes((i % 256) for i in range(256 * 1024 + 1))
    c = o.compress(t, 6, o.FMT_DEFLATE)
    assert c.count(b"\x00\x00\xff\xff") >= 1
    assert zlib.decompress(c, -15) == t
    s = o.compress_raw(b"hello hello hello", 6, o.FLUSH_SYNC)
    assert s.endswith(b"\x00\x00\xff\xff")


def test_batch_semantics(kat):
    # tests/batch_test.rs
    inputs = [expand(d) for d in kat["batch_roundtrip_inputs"]]
    flat = np.frombuffer(b"".join(inputs), dtype=np.uint8)
    in_off = np.zeros(len(inputs) + 1, dtype=np.uint64)
    in_off[1:] = np.cumsum([len(x) for x in inputs])
    out, out_off, sizes, status = o.batch_compress(flat, in_off, o.FMT_DEFLATE, 6, threads=4)
    assert (status == 0).all()
    comp = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])].tobytes() for i in range(len(inputs))]
    cflat = np.frombuffer(b"".join(comp), dtype=np.uint8)
    c_off = np.zeros(len(inputs) + 1, dtype=np.uint64)
    c_off[1:] = np.cumsum([len(x) for x in comp])
    d_off = in_off.copy()
    dout, dsizes, dstatus = o.batch_decompress(cflat, c_off, d_off, o.FMT_DEFLATE, threads=4)
    assert (dstatus == 0).all()
    for i, t in enumerate(inputs):
        assert dout[int(d_off[i]):int(d_off[i]) + int(dsizes[i])].tobytes() == t
    # empty input compresses to a non-empty stream that inflates into zero bytes
    e = o.compress(b"", 6, o.FMT_DEFLATE)
    assert e and o.decompress(e, 0) == (o.SUCCESS, len(e), b"")
    v = kat["batch_insufficient"]
    c = o.compress(expand(v["input"]), 6, o.FMT_DEFLATE)
    assert o.decompress(c, v["max_out"])[0] != o.SUCCESS


def test_incompressible_fails_like_reference():
    # no stored-block fallback above level 0 (src/compress/mod.rs:1475-1496):
    # random bytes do not fit deflate_compress_bound and report InsufficientSpace
    assert o.compress(os.urandom(65536), 6, o.FMT_DEFLATE) is None


def test_near_optimal_levels():
    # levels 10-12 (src/compress/mod.rs:1586-1773): streams valid for zlib and for the oracle's
    # own inflater, no larger than level 9 on compressible data, and the reference's own
    # ordering test (tests/unit_tests.rs:112-125)
    import corpus
    datas = [corpus.json_like(300000, 31), corpus.text(300000, 32), corpus.mixed(300000, 33), b"",
             b"a", b"abcabcabc" * 3000, bytes(70000), corpus.text(262144 + 5000, 34)]
    for lvl in (10, 11, 12):
        for d in datas:
            c = o.compress(d, lvl)
            assert c is not None
            assert zlib.decompress(c, -15) == d
            assert o.decompress(c, len(d), o.FMT_DEFLATE) == (o.SUCCESS, len(c), d)
    for d in datas[:3]:
        assert len(o.compress(d, 12)) <= len(o.compress(d, 9))
    a = b"a" * 10000
    assert len(o.compress(a, 0)) > 10000
    assert len(o.compress(a, 12)) <= len(o.compress(a, 1)) < len(o.compress(a, 0))
    for fmt, wb in ((o.FMT_ZLIB, 15), (o.FMT_GZIP, 31)):
        assert zlib.decompress(o.compress(datas[0], 12, fmt), wb) == datas[0]


def test_dynamic_headers_carry_make_huffman_code_lengths():
    # huffman_comp.rs:125-155 through the test hook: the code lengths in the oracle's own dynamic headers are the
    # ones make_huffman_code gives for the frequencies counted from the block bodies by the tests' DEFLATE walker
    import corpus
    import deflate_walk
    for level in (2, 6, 9):
        stream = o.compress(corpus.text(150000, 5) + corpus.mixed(80000, 6), level)
        blocks = [b for b in deflate_walk.walk(stream) if b["type"] == 2]
        assert blocks
        for b in blocks:
            for key, fkey, nsym, max_len in (("litlen_lens", "litlen_freq", 288, 14), ("offset_lens", "offset_freq", 32, 15)):
                want, _ = o.make_huffman_code(b[fkey][:nsym], max_len)
                got = list(b[key]) + [0] * (nsym - len(b[key]))
                assert got == [int(x) for x in want]
    # the degenerate codes of :133-145: no symbol, one symbol
    lens, codes = o.make_huffman_code([0] * 32, 15)
    assert list(lens[:2]) == [1, 1] and not lens[2:].any()
    lens, codes = o.make_huffman_code([0] * 7 + [9] + [0] * 24, 15)
    assert lens[0] == 1 and lens[7] == 1 and int(lens.sum()) == 2
