"""GPU inflate (raw / zlib / gzip) through the C ABI vs the oracle."""
import zlib

import numpy as np
import pytest

import corpus
import handmade
import oracle_lib as o
import rsx_pkg
from golden.expand import expand

pytestmark = pytest.mark.gpu

WBITS = {0: -15, 1: 15, 2: 31}


@pytest.fixture(scope="module")
def rsx():
    return rsx_pkg.load()


@pytest.fixture(scope="module")
def eng(rsx):
    return rsx.default_engine()


def check_against_oracle(eng, streams, caps, fmt):
    got, status = eng.batch_decompress(streams, caps, fmt)
    for i, s in enumerate(streams):
        r, _, want = o.decompress(s, caps[i], fmt)
        assert int(status[i]) == r, (i, int(status[i]), r)
        if r == o.SUCCESS:
            assert got[i] == want, i
        else:
            assert got[i] is None


def test_reference_kat_streams(rsx, eng, kat):
    streams = [bytes.fromhex(v["stream_hex"]) for v in kat["inflate"]]
    wants = [expand(v["output"]) for v in kat["inflate"]]
    got, status = eng.batch_decompress(streams, [len(w) for w in wants])
    assert list(status) == [0, 0] and got == wants
    bad = [bytes.fromhex(v["stream_hex"]) for v in kat["inflate_failures"]]
    got, status = eng.batch_decompress(bad, [v["max_out"] for v in kat["inflate_failures"]])
    assert got == [None]


def test_batch_semantics_like_reference(rsx, kat):
    # tests/batch_test.rs with streams produced by the oracle's level-6 compressor
    inputs = [expand(d) for d in kat["batch_roundtrip_inputs"]]
    comp = [o.compress(t, 6) for t in inputs]
    dec = rsx.BatchDecompressor()
    assert dec.decompress_batch(comp, [len(t) for t in inputs]) == inputs
    assert dec.decompress_batch([], []) == []
    v = kat["batch_empty_input"]
    ins = [expand(d) for d in v["inputs"]]
    comp = [o.compress(t, 6) for t in ins]
    assert dec.decompress_batch(comp, v["max_out"]) == ins
    v = kat["batch_insufficient"]
    c = o.compress(expand(v["input"]), 6)
    assert dec.decompress_batch([c], [v["max_out"]]) == [None]
    assert dec.decompress_batch([bytes([0, 1, 2, 3, 4, 5])], [100]) == [None]


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_oracle_streams_all_levels(eng, fmt):
    datas = [corpus.text(300000, 1), corpus.mixed(400000, 2), corpus.json_like(200000, 3),
             bytes(100000), b"a", b"ab" * 5000, corpus.text(5000, 4)]
    streams, caps = [], []
    for lvl in (1, 2, 6, 9):
        for d in datas:
            streams.append(o.compress(d, lvl, fmt))
            caps.append(len(d))
    check_against_oracle(eng, streams, caps, fmt)


def test_zlib_streams_including_stored_and_fixed(eng):
    datas = [corpus.text(200000, 5), corpus.random_bytes(150000, 6), corpus.mixed(250000, 7)]
    streams, caps = [], []
    for d in datas:
        for lvl in (0, 1, 6, 9):
            streams.append(zlib.compress(d, lvl))
            caps.append(len(d) + 100)
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
        streams.append(co.compress(d) + co.flush())
        caps.append(len(d))
    check_against_oracle(eng, streams, caps, 1)


def test_overlapping_copies_every_small_offset(eng):
    # tests/offset_tests.rs: period-N patterns force matches with offset < length
    streams, caps = [], []
    for period in list(range(1, 40)) + [63, 64, 65, 127, 258, 259, 1000]:
        for n in (1000, 100000):
            d = corpus.periodic(period, n)
            for lvl in (6, 1):
                c = o.compress(d, lvl)
                if c is not None:  # a short incompressible pattern fails like the reference
                    streams.append(c)
                    caps.append(n)
    check_against_oracle(eng, streams, caps, 0)


def test_far_matches_and_window_edges(eng):
    # matches at distance 32768 and around the LDS ring boundary
    rs = np.random.RandomState(11)
    block = rs.randint(0, 256, size=32768, dtype=np.uint8).tobytes()
    d = block + block + block[:1000] + bytes(70000) + block
    streams = [zlib.compress(d, 9)[2:-4], o.compress(d, 9)]
    check_against_oracle(eng, streams, [len(d)] * 2, 0)


def test_stored_block_longer_than_the_rest_of_the_stream(eng):
    # zlib sends the first pass over a random block as a stored block and everything behind it as matches
    # at one distance: the stored block is longer than the rest of the body, and the Huffman block behind it
    # is a run of identical tokens in which every bit offset is a plausible token start
    streams, caps = [], []
    for period, seed in ((20000, 1), (20000, 2), (16500, 3), (30000, 4)):
        rs = np.random.RandomState(seed)
        block = rs.randint(0, 256, size=period, dtype=np.uint8).tobytes()
        d = (block * (400000 // period + 1))[:400000]
        streams += [zlib.compress(d, 6)] * 24
        caps += [len(d)] * 24
    for _ in range(3):
        check_against_oracle(eng, streams, caps, 1)
        assert eng.last_fallbacks() == 0   # (decoded once: no stream was handed to the one-kernel decoder)


def test_streams_that_compress_a_thousandfold(eng):
    # few, long matches: such a stream owns few token rows and gets by with short sub-sequences
    streams, caps = [], []
    for d in (bytes(1 << 20), b"\xAB" * 700001, b"abcd" * 200000, corpus.periodic(100, 1 << 20),
              corpus.periodic(258, 500000), corpus.text(3000, 9) * 300):
        for lvl in (1, 6, 9):
            streams.append(zlib.compress(d, lvl))
            caps.append(len(d))
    check_against_oracle(eng, streams, caps, 1)
    assert eng.last_fallbacks() == 0


def test_valid_streams_are_decoded_once(eng):
    # the one-kernel decoder behind the two-kernel one is for damaged streams and small slots: a batch of
    # valid streams of every kind must not need it (a stream that does is right, but slow)
    datas = [corpus.text(300000, 21), corpus.mixed(400000, 22), corpus.json_like(200000, 23), corpus.random_bytes(70000, 24),
             bytes(100000), b"a", b"", b"ab" * 5000, corpus.text(5000, 25)]
    for fmt in (0, 1, 2):
        streams, caps = [], []
        for lvl in (1, 6, 9):
            for d in datas:
                z = o.compress(d, lvl, fmt)
                if z is not None:   # (an input the reference's compressor refuses)
                    streams.append(z)
                    caps.append(len(d))
        if fmt:
            for d in datas:
                for lvl in (0, 1, 6, 9):
                    co = zlib.compressobj(lvl, zlib.DEFLATED, WBITS[fmt])
                    streams.append(co.compress(d) + co.flush())
                    caps.append(len(d) + 7)
        check_against_oracle(eng, streams, caps, fmt)
        assert eng.last_fallbacks() == 0, fmt
    # ... and the counter counts: one damaged stream, one slot that is too small
    good = zlib.compress(datas[0], 6)
    bad = bytearray(good)
    bad[len(bad) // 2] ^= 0x55
    check_against_oracle(eng, [good, bytes(bad), good], [len(datas[0]), len(datas[0]), 1000], 1)
    assert eng.last_fallbacks() == 2


def test_streams_of_short_blocks_share_token_rows(eng):
    # zlib at every memLevel (blocks of 127 symbols at memLevel 1, of 16383 at memLevel 8), every strategy, with and
    # without a flush every 20 KB: rounds end where blocks end, so streams of short blocks and streams of nothing
    # but literals (Z_HUFFMAN_ONLY) fill a few columns of a round's token rows.  Rounds share rows (and a stream
    # that still runs out borrows from the batch's pool): none of these streams may need the one-kernel decoder.
    datas = [corpus.text(300000, 31), corpus.mixed(300000, 32), corpus.json_like(300000, 33), corpus.text(8000, 35)]
    for mem in (1, 2, 4, 6, 8, 9):
        for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
            for flush in (None, zlib.Z_SYNC_FLUSH, zlib.Z_BLOCK):
                if mem not in (1, 8) and flush == zlib.Z_BLOCK:
                    continue
                streams, caps = [], []
                for d in datas:
                    for lvl in (1, 6):
                        co = zlib.compressobj(lvl, zlib.DEFLATED, 15, mem, strat)
                        parts = []
                        step = len(d) if flush is None else 20000
                        for at in range(0, len(d), step):
                            parts.append(co.compress(d[at:at + step]))
                            if flush is not None:
                                parts.append(co.flush(flush))
                        parts.append(co.flush())
                        streams.append(b"".join(parts))
                        caps.append(len(d))
                got, status = eng.batch_decompress(streams, caps, 1)
                assert [int(x) for x in status] == [0] * len(streams), (mem, strat, flush)
                assert got == [d for d in datas for _ in (1, 6)], (mem, strat, flush)
                assert eng.last_fallbacks() == 0, (mem, strat, flush)
    # a batch made of nothing but such streams (the pool is a quarter of the batch's own rows)
    streams = []
    for k in range(64):
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 1 + k % 3, zlib.Z_HUFFMAN_ONLY if k % 2 else zlib.Z_DEFAULT_STRATEGY)
        streams.append(co.compress(datas[k % 3]) + co.flush())
    got, status = eng.batch_decompress(streams, [300000] * 64, 1)
    assert not any(status) and got == [datas[k % 3] for k in range(64)]
    assert eng.last_fallbacks() == 0


def test_error_codes_match_oracle(eng):
    t = corpus.text(50000, 8)
    c = o.compress(t, 6)
    g = o.compress(t, 6, 2)
    z = o.compress(t, 6, 1)
    bad_crc = bytearray(g); bad_crc[-6] ^= 1
    bad_isize = bytearray(g); bad_isize[-1] ^= 1
    bad_adler = bytearray(z); bad_adler[-1] ^= 1
    cases = [
        (c, len(t) - 1, 0), (c, 10, 0), (c[:len(c) // 2], len(t), 0), (c[:1], len(t), 0),
        (b"", 10, 0), (b"\x07", 10, 0), (b"\x01\x05\x00\x00\x00", 10, 0),
        (b"\x01\x05\x00\xfa\xffabc", 10, 0), (b"\x01\x05\x00\xfa\xffabcde", 4, 0),
        (bytes([0x03, 0x02, 0x00, 0x00]), 100, 0),
        (bytes(bad_crc), len(t), 2), (bytes(bad_isize), len(t), 2), (bytes(bad_adler), len(t), 1),
        (b"\x78\x9d" + z[2:], len(t), 1), (z[:5], len(t), 1), (g[:17], len(t), 2),
        (b"\x1f\x8b\x09" + g[3:], len(t), 2), (g, len(t) - 1, 2), (z, len(t) + 5, 1),
    ]
    rs = np.random.RandomState(5)
    for k in range(40):  # random corruption of a valid stream
        b = bytearray(c)
        for _ in range(3):
            b[rs.randint(0, len(b))] ^= 1 << rs.randint(0, 8)
        cases.append((bytes(b), len(t), 0))
    for fmt in (0, 1, 2):
        sub = [(s, cap) for s, cap, f in cases if f == fmt]
        if sub:
            check_against_oracle(eng, [s for s, _ in sub], [cap for _, cap in sub], fmt)


def test_gzip_header_fields(eng):
    import gzip, io
    t = corpus.text(20000, 9)
    buf = io.BytesIO()
    with gzip.GzipFile(filename="name.txt", mode="wb", fileobj=buf, mtime=0) as f:
        f.write(t)
    check_against_oracle(eng, [buf.getvalue()], [len(t)], 2)


@pytest.mark.parametrize("seed", [2024, 7, 99, 12345])
def test_corrupted_streams_agree_with_oracle(eng, seed):
    """Six hundred damaged streams per seed (bit flips, cuts, spliced tails, short slots) in all three
    containers: the status code of every item, and the bytes of every item that still decodes, must be
    the oracle's.  Damage lands in headers, code-length tables, token data and trailers alike."""
    rs = np.random.RandomState(seed)
    texts = [corpus.text(40000, 50), corpus.mixed(30000, 51), bytes(5000), corpus.text(300, 52),
             rs.randint(0, 256, size=3000, dtype=np.uint8).tobytes()]
    for fmt in (0, 1, 2):
        good = []
        for k, t in enumerate(texts):
            level = 0 if k == 4 else (1, 6, 9)[k % 3]   # incompressible bytes go in as stored blocks
            good.append((o.compress(t, level, fmt), len(t)))
        streams, caps = [], []
        for s, n in good:
            for _ in range(40):
                b = bytearray(s)
                kind = rs.randint(0, 5)
                if kind == 0:        # a few bit flips anywhere
                    for _ in range(rs.randint(1, 4)):
                        b[rs.randint(0, len(b))] ^= 1 << rs.randint(0, 8)
                elif kind == 1:      # damage near the front: block header / code lengths
                    b[rs.randint(0, min(len(b), 80))] ^= 1 << rs.randint(0, 8)
                elif kind == 2:      # cut
                    b = b[:rs.randint(0, len(b))]
                elif kind == 3:      # overwrite the tail with noise
                    at = rs.randint(0, len(b))
                    b[at:] = rs.randint(0, 256, size=len(b) - at, dtype=np.uint8).tobytes()
                else:                # intact stream, slot too small or larger than needed
                    pass
                streams.append(bytes(b))
                caps.append(n if kind != 4 else int(rs.choice([0, n // 2, n - 1, n, n + 7])))
        check_against_oracle(eng, streams, caps, fmt)


def test_matches_across_copy_step_boundaries(eng):
    """Long matches back to back (a 3000-byte random block repeated, so almost every token is a
    258-byte match at distance 3000) cross the 2048-byte steps of the copy phase at every phase;
    a few literals in between shift the alignment.  Also short periods, where one match feeds the
    next inside a step."""
    rs = np.random.RandomState(77)
    block = rs.randint(0, 256, size=3000, dtype=np.uint8).tobytes()
    datas = []
    for shift in (0, 1, 2, 3, 7, 100, 257, 2047):
        d = bytes(shift) + b"".join(block + bytes([k & 0xFF]) * (k % 3) for k in range(60))
        datas.append(d)
    datas.append(block[:5] * 40000)            # period 5, 200000 bytes: overlapping run-length copies
    datas.append((block[:2049]) * 100)         # period one byte longer than a copy step
    datas.append((block[:2047]) * 100)
    streams = [zlib.compress(d, 9)[2:-4] for d in datas] + [o.compress(d, 6) for d in datas]
    check_against_oracle(eng, streams, [len(d) for d in datas] * 2, 0)


def test_long_codewords_use_subtables(eng):
    """Skewed symbol statistics give codewords up to 15 bits, longer than the root index of the litlen
    table (10 bits) and of the offset table (8 bits): geometric byte distributions for the literals,
    and copies from one favourite distance with rare copies from all the others for the offsets."""
    rs = np.random.RandomState(4242)
    datas = []
    for ratio in (0.5, 0.62, 0.75):
        p = ratio ** np.arange(256, dtype=np.float64)
        p /= p.sum()
        datas.append(rs.choice(256, size=200000, p=p).astype(np.uint8).tobytes())
    for favourite in (5, 300, 20000):
        out = bytearray(rs.randint(0, 256, size=33000, dtype=np.uint8).tobytes())
        while len(out) < 400000:
            rare = rs.randint(0, 400) == 0
            dist = int(rs.randint(1, 32768)) if rare else favourite
            n = int(rs.randint(3, 7))
            start = len(out) - dist
            for k in range(n):
                out.append(out[start + k])
            if rs.randint(0, 3) == 0:
                out.append(int(rs.randint(0, 256)))
        datas.append(bytes(out))
    streams, caps = [], []
    for d in datas:
        for lvl in (1, 6, 9):
            streams.append(zlib.compress(d, lvl)[2:-4])
            caps.append(len(d))
        c = o.compress(d, 9)
        if c is not None:
            streams.append(c)
            caps.append(len(d))
    check_against_oracle(eng, streams, caps, 0)


def test_dynamic_headers_built_by_hand(eng):
    """Block headers that no compressor writes (tests/handmade.py): code lengths sent one by one with the
    longest precode codewords, runs of every kind at every place, a repeat as the first token, and
    truncations all along."""
    streams, caps, text = handmade.header_cases()
    check_against_oracle(eng, streams, caps, 0)
    got, status = eng.batch_decompress([streams[0]], [len(text)], 0)
    assert int(status[0]) == 0 and got[0] == text


def test_gzip_files_with_several_members(rsx, eng):
    """rsx_batch_decompress_members: every member of every item, appended in its slot; checked against
    Python's gzip module (which reads multi-member files) and, member by member, against the oracle."""
    import gzip
    parts = [corpus.text(300000, 71), corpus.mixed(200000, 72), b"", corpus.json_like(70000, 73), b"x"]
    gz = [gzip.compress(p, 6, mtime=0) for p in parts]
    for p, g in zip(parts, gz):
        assert o.decompress(g, len(p), o.FMT_GZIP)[0] == o.SUCCESS
    items = [gz[0],                                   # one member
             gz[0] + gz[1],                           # two
             b"".join(gz),                            # five, with an empty one in the middle
             gz[3] + bytes(512),                      # zero padding behind the member
             gz[1] + b"trailing text that is no member",
             gz[0] + gz[1][:-9],                      # second member cut short
             gz[4] * 40,                              # forty one-byte members
             gz[0] + gz[1][:100] + b"\xff" * 50 + gz[1][150:]]   # second member damaged
    want = [parts[0], parts[0] + parts[1], b"".join(parts), parts[3], None, None, b"x" * 40, None]
    caps = [700000] * len(items)
    got, st = eng.batch_decompress_members(items, caps)
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, i
        assert (st[i] == 0) == (w is not None), (i, st[i])
        if w is not None:
            assert gzip.decompress(items[i]) == w
    assert st[4] == rsx.BAD_DATA
    # a slot that holds the first member but not the second
    got, st = eng.batch_decompress_members([gz[0] + gz[1]], [len(parts[0]) + 10])
    assert got == [None] and st[0] == rsx.INSUFFICIENT_SPACE
    # the single-member entry point stops behind the first member, like the reference
    got, st = eng.batch_decompress([gz[0] + gz[1]], [700000], rsx.FORMAT_GZIP)
    assert got == [parts[0]] and st[0] == 0


# ---- few large streams: cut at sync-flush joints and decoded part by part ------------------

def test_large_streams_are_cut_into_parts(rsx, eng):
    # src/compress/mod.rs:699-772: inputs above 256 KiB are compressed in independent 256 KiB chunks joined by
    # sync-flush markers; the decoder starts one part per joint it finds (at most one per 32 KiB of input)
    data = corpus.text(3 << 20, 11) + corpus.mixed(1 << 20, 12) + corpus.json_like(1 << 20, 13)
    for fmt in (rsx.FORMAT_DEFLATE, rsx.FORMAT_ZLIB, rsx.FORMAT_GZIP):
        stream = o.compress(data, 6, fmt)
        got, status = eng.batch_decompress([stream], [len(data)], fmt)
        assert list(status) == [0] and got[0] == data
        assert 18 <= eng.last_parts() <= (len(data) + (1 << 18) - 1) >> 18, eng.last_parts()   # one part per chunk
        assert eng.last_whole() == 0
    # the engine's own compressor produces the same joints
    comp, cstatus = eng.batch_compress([data], rsx.FORMAT_GZIP, 6)
    assert list(cstatus) == [0]
    got, status = eng.batch_decompress(comp, [len(data)], rsx.FORMAT_GZIP)
    assert list(status) == [0] and got[0] == data and eng.last_parts() > 8 and eng.last_whole() == 0


def test_parts_with_history_across_the_joint(rsx, eng):
    # zlib's Z_SYNC_FLUSH keeps the window: matches behind the joint reach in front of it, so the parts cannot
    # be expanded side by side (their tokens are expanded by one workgroup per stream, part after part);
    # Z_FULL_FLUSH forgets it.  Either way the bytes are zlib's.
    data = corpus.text(2 << 20, 21)
    for flush in (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        stream = b""
        for at in range(0, len(data), 200000):
            stream += co.compress(data[at:at + 200000]) + co.flush(flush)
        stream += co.flush()
        got, status = eng.batch_decompress([stream, stream], [len(data), len(data) + 5], rsx.FORMAT_DEFLATE)
        assert list(status) == [0, 0] and got == [data, data]
        assert eng.last_parts() >= 2 and eng.last_whole() == (2 if flush == zlib.Z_SYNC_FLUSH else 0)


def test_parts_and_false_joints(rsx, eng):
    # the marker bytes inside stored data and inside a gzip header field are not joints; a stream cut short,
    # a slot that is too small and damaged parts report what the oracle reports
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    marked = bytearray(noise)
    for at in range(1000, len(marked) - 8, 50000):
        marked[at:at + 4] = b"\x00\x00\xff\xff"
    marked = bytes(marked)
    text = corpus.text(1 << 20, 31)
    mixed = text + marked + text
    co, c0 = zlib.compressobj(6, zlib.DEFLATED, 31), zlib.compressobj(0, zlib.DEFLATED, 31)
    streams = [c0.compress(marked) + c0.flush(), co.compress(mixed) + co.flush()]   # stored blocks hold the noise
    caps = [len(marked), len(mixed)]
    good = o.compress(text + text, 6, o.FMT_GZIP)
    streams += [good[:len(good) // 2], good, good[:-9] + b"\x00" + good[-8:], good]
    caps += [2 * len(text), 2 * len(text) - 1, 2 * len(text), 2 * len(text) + 100]
    damaged = bytearray(good)
    damaged[len(good) // 3] ^= 0x55
    streams.append(bytes(damaged))
    caps.append(2 * len(text))
    check_against_oracle(eng, streams, caps, rsx.FORMAT_GZIP)
    assert eng.last_parts() > len(streams)


def test_final_block_that_ends_at_a_joint_ends_the_stream(rsx, eng):
    # Go's flate.Writer.Close() ends a stream with an empty stored FINAL block, 01 00 00 FF FF, which holds the
    # marker the cut search looks for.  Bytes behind it that happen to decode (a second stream) are not part of
    # the item: the reference stops at the final block (src/decompress/mod.rs:246-249) and so must a stream that
    # is cut into parts there.
    rng = np.random.default_rng(17)
    noise = bytearray(rng.integers(0, 256, 1200000, dtype=np.uint8).tobytes())
    at = noise.find(b"\x00\x00\xff\xff")
    while at >= 0:   # (the only markers in the item are the ones placed below)
        noise[at] = 1
        at = noise.find(b"\x00\x00\xff\xff")
    noise = bytes(noise)

    def stored(b, final=0):
        return bytes([final]) + len(b).to_bytes(2, "little") + (len(b) ^ 0xFFFF).to_bytes(2, "little") + b

    head = noise[:1100000]
    a = b"".join(stored(head[k:k + 65535]) for k in range(0, len(head), 65535))
    a += stored(b"")                       # a full flush
    a += stored(noise[1100000:1140000])    # more than a search tile of stored bytes behind it ...
    a += stored(b"", 1)                    # ... so that this marker is the first of its tile
    a_data = noise[:1140000]
    text = corpus.text(300000, 77)
    b = zlib.compress(text, 6)[2:-4]
    check_against_oracle(eng, [a + b, a], [len(a_data) + len(text) + 100, len(a_data)], rsx.FORMAT_DEFLATE)
    got, status = eng.batch_decompress([a + b], [len(a_data) + len(text) + 100], rsx.FORMAT_DEFLATE)
    assert list(status) == [0] and got[0] == a_data
    assert eng.last_parts() >= 2
    # with a trailer behind the final block the trailer is found where the stream ends
    z = b"\x78\x9c" + a + zlib.adler32(a_data).to_bytes(4, "big")
    check_against_oracle(eng, [z, z + b], [len(a_data), len(a_data) + 7], rsx.FORMAT_ZLIB)


def test_parts_are_off_for_short_streams(rsx, eng):
    # (streams that average less than 256 KiB of input are not searched for joints)
    data = [corpus.text(300000, 40 + i) for i in range(4)]
    streams = [o.compress(d, 1) for d in data] * 275
    got, status = eng.batch_decompress(streams, [300000] * len(streams))
    assert not any(status) and got[:4] == data[:4] and got[-4:] == data[:4]
    assert eng.last_parts() == 0


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_parts_random_flush_patterns(rsx, eng, seed):
    """Streams of one to four MiB with flushes of every kind at random places (full flushes and chunk joints
    can be cut, sync flushes keep their history, block flushes are no joints at all), stored stretches, in all
    three containers, some cut short or with a slot one byte too small: status and bytes must be the oracle's."""
    rs = np.random.RandomState(seed)
    kinds = [zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH, zlib.Z_FULL_FLUSH]
    if hasattr(zlib, "Z_BLOCK"):
        kinds.append(zlib.Z_BLOCK)
    for fmt in (0, 1, 2):
        streams, caps = [], []
        for k in range(4):
            size = int(rs.randint(1 << 20, 4 << 20))
            pieces = [corpus.text(size // 2, 100 * seed + k), corpus.random_bytes(size // 8, seed + k),
                      corpus.json_like(size - size // 2 - size // 8, 7 * seed + k)]
            data = b"".join(pieces)
            co = zlib.compressobj(int(rs.choice([1, 6, 9])), zlib.DEFLATED, WBITS[fmt])
            out, at = [], 0
            while at < len(data):
                n = int(rs.randint(1000, 400000))
                out.append(co.compress(data[at:at + n]))
                out.append(co.flush(int(rs.choice(kinds))) if rs.randint(0, 3) else b"")
                at += n
            out.append(co.flush())
            s = b"".join(out)
            own = eng.batch_compress([data], fmt, 6)[0][0]   # (chunk joints; the incompressible chunk as stored blocks)
            streams += [s, s[:int(rs.randint(1, len(s)))], s, own]
            caps += [len(data), len(data), len(data) - 1, len(data)]
        check_against_oracle(eng, streams, caps, fmt)
        assert eng.last_parts() > len(streams)
