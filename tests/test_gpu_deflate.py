"""GPU deflate (raw / zlib / gzip) through the C ABI: every stream must be valid for
its format and inflate -- with the oracle, with zlib and with the GPU inflater --
to exactly the input; sizes stay close to the oracle's at the same level."""
import zlib

import numpy as np
import pytest

import corpus
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


def roundtrip(eng, datas, fmt, level):
    comp, status = eng.batch_compress(datas, fmt, level)
    assert list(status) == [0] * len(datas), list(status)
    for d, c in zip(datas, comp):
        assert len(c) <= o.compress_bound(fmt, len(d))
        assert zlib.decompress(c, WBITS[fmt]) == d
        assert o.decompress(c, len(d), fmt) == (o.SUCCESS, len(c), d)
    back, st = eng.batch_decompress(comp, [len(d) for d in datas], fmt)
    assert list(st) == [0] * len(datas) and back == list(datas)
    return comp


def test_reference_batch_tests(rsx, kat):
    # tests/batch_test.rs through the mirrored API
    inputs = [expand(d) for d in kat["batch_roundtrip_inputs"]]
    comp = rsx.BatchCompressor(6).compress_batch(inputs)
    assert len(comp) == len(inputs) and all(len(c) > 0 for c in comp)
    assert rsx.BatchDecompressor().decompress_batch(comp, [len(t) for t in inputs]) == inputs
    assert rsx.BatchCompressor(6).compress_batch([]) == []
    v = kat["batch_empty_input"]
    ins = [expand(d) for d in v["inputs"]]
    comp = rsx.BatchCompressor(6).compress_batch(ins)
    assert len(comp[0]) > 0
    assert rsx.BatchDecompressor().decompress_batch(comp, v["max_out"]) == ins
    v = kat["batch_insufficient"]
    comp = rsx.BatchCompressor(6).compress_batch([expand(v["input"])])
    assert rsx.BatchDecompressor().decompress_batch(comp, [v["max_out"]]) == [None]


@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("level", [0, 1, 4, 6, 9, 10, 12])
def test_roundtrip_formats_levels(eng, fmt, level):
    datas = [corpus.text(300000, 1), corpus.mixed(400000, 2), corpus.json_like(100000, 3), b"",
             b"a", b"abc", b"ab" * 3000, bytes(70000), corpus.random_bytes(5000, 4),
             corpus.text(262144, 5), corpus.text(262145, 6), corpus.text(511, 7),
             corpus.text(512, 8), corpus.text(513, 9), corpus.text(1025, 10)]
    roundtrip(eng, datas, fmt, level)


def test_parallel_test_shapes(eng):
    # tests/parallel_test.rs: inputs that exercise the >256 KiB chunked path
    datas = [bytes(i % 256 for i in range(1 << 20)), bytes(i % 256 for i in range(256 * 1024 + 1)),
             bytes((i * 3) % 251 for i in range(600000)), bytes((i * 7) % 251 for i in range(600000))]
    comp = roundtrip(eng, datas, 0, 6)
    assert comp[0].count(b"\x00\x00\xff\xff") >= 3  # sync-flush joints between the four chunks
    roundtrip(eng, datas[2:], 1, 6)
    roundtrip(eng, datas[2:], 2, 6)


def test_periodic_patterns(eng):
    # tests/offset_tests.rs shapes
    datas = [corpus.periodic(p, n) for p in list(range(1, 34)) + [63, 64, 65, 258, 259, 1000]
             for n in (1000, 100000)]
    roundtrip(eng, datas, 0, 6)
    roundtrip(eng, datas[::5], 0, 1)


def test_incompressible_uses_stored_blocks(eng):
    # the reference fails here (no stored fallback, src/compress/mod.rs:1475-1496); the
    # engine emits stored blocks and fits the bound
    d = corpus.random_bytes(65536, 12)
    assert o.compress(d, 6) is None
    comp = roundtrip(eng, [d, corpus.random_bytes(200000, 13)], 0, 6)
    assert len(comp[0]) <= len(d) + 16


def test_level_ordering_and_ratio_vs_oracle(eng):
    # tests/unit_tests.rs:112-125: L0 > L1 >= L12 on 10000 x 'a'
    d = b"a" * 10000
    sizes = {lvl: len(eng.batch_compress([d], 0, lvl)[0][0]) for lvl in (0, 1, 6, 12)}
    assert sizes[0] > sizes[1] >= sizes[12]
    # sizes against the oracle at the same level: within 2 % on text, mixed and JSON-like data at levels 6
    # and 9, except JSON at level 6, where the row table of levels 1-6 offers twelve candidates per 4-byte
    # string against the 35 steps of the reference's chain (8 % larger; level 9 walks chains: 2 % smaller)
    sets = {"text": [corpus.text(1 << 20, s) for s in range(2)], "mixed": [corpus.mixed(1 << 20, 7)],
            "json": [corpus.json_like(1 << 20, 21)]}
    for level in (6, 9):
        for kind, datas in sets.items():
            comp, st = eng.batch_compress(datas, 0, level)
            assert list(st) == [0] * len(datas)
            got = sum(len(c) for c in comp)
            ref = sum(len(o.compress(d, level)) for d in datas)
            limit = 1.09 if (kind, level) == ("json", 6) else 1.02
            assert got <= limit * ref, (kind, level, got, ref)


def test_code_lengths_are_the_reference_algorithms(eng):
    # make_huffman_code (src/compress/huffman_comp.rs:125-155): the code lengths in the dynamic headers the engine
    # writes are the ones the oracle's restatement computes for the same symbol frequencies (the frequencies are
    # counted from the block bodies by an independent DEFLATE walker), length limiting included: the skewed
    # buffer has symbols 2^-18 as frequent as the commonest, whose unlimited codewords would be longer than 14 bits
    import deflate_walk
    rs = np.random.RandomState(5)
    p = 0.55 ** np.arange(40, dtype=np.float64)
    skewed = rs.choice(40, size=60000, p=p / p.sum()).astype(np.uint8).tobytes()
    datas = [corpus.text(200000, 71), corpus.mixed(150000, 72), corpus.json_like(120000, 73), skewed]
    checked = limited = 0
    for level in (2, 6, 9):
        comp, st = eng.batch_compress(datas, 0, level)
        for c in comp:
            for blk in deflate_walk.walk(c):
                if blk["type"] != 2:
                    continue
                for lens_key, freq_key, nsym, max_len in (("litlen_lens", "litlen_freq", 288, 14),
                                                          ("offset_lens", "offset_freq", 32, 15)):
                    want, _ = o.make_huffman_code(blk[freq_key][:nsym], max_len)
                    got = list(blk[lens_key]) + [0] * (nsym - len(blk[lens_key]))
                    assert got == [int(x) for x in want], (level, lens_key)
                    limited += max(got) == max_len
                checked += 1
    assert checked >= 20 and limited >= 1


def test_level_one_uses_the_static_code(eng):
    # src/compress/mod.rs:1499-1583: level 1 codes every block with the fixed code of RFC 1951 (BTYPE = 01) behind
    # a single-probe match finder (matchfinder.rs:1109-1232); blocks that do not pay are still stored
    datas = [corpus.text(1 << 20, 61), corpus.mixed(1 << 20, 62), corpus.json_like(1 << 20, 63), corpus.text(3000, 64)]
    comp = roundtrip(eng, datas, 0, 1)
    for d, c in zip(datas, comp):
        assert (c[0] >> 1) & 3 in (1, 0)            # first block: static (or stored)
        ref = o.compress(d, 1)
        assert (ref[0] >> 1) & 3 == 1
        assert len(c) <= 1.06 * len(ref), (len(c), len(ref))
    assert (comp[0][0] >> 1) & 3 == 1 and (comp[3][0] >> 1) & 3 == 1
    roundtrip(eng, datas, 1, 1)
    roundtrip(eng, datas, 2, 1)


def test_cost_based_levels(eng):
    # levels 10-12 (src/compress/mod.rs:1586-1773): valid streams in every format, no larger
    # than the lazy parse with the deepest search, and within 3 % of the oracle's restatement
    # of the reference's near-optimal parse at the same level (the engine prices every prefix
    # of every recorded match, the reference only whole matches: it is usually smaller)
    datas = [corpus.json_like(1 << 20, 21), corpus.text(700000, 22), corpus.mixed(300000, 23),
             corpus.text(8191, 24), corpus.text(8193, 25), b"abcabcabc" * 5000, corpus.random_bytes(70000, 26)]
    l9, _ = eng.batch_compress(datas, 0, 9)
    for level in (10, 11, 12):
        comp = roundtrip(eng, datas, 0, level)
        for d, c, c9 in zip(datas[:3], comp, l9):
            ref = o.compress(d, level)
            assert len(c) <= 1.03 * len(ref), (level, len(c), len(ref))
            if level == 12:
                assert len(c) <= len(c9), (len(c), len(c9))
    roundtrip(eng, datas[:3], 1, 12)
    roundtrip(eng, datas[:3], 2, 10)


def test_api_surface(rsx):
    c = rsx.Compressor(6)
    d = corpus.text(50000, 3)
    for comp, dec in ((c.compress_deflate, "decompress_deflate"), (c.compress_zlib, "decompress_zlib"),
                      (c.compress_gzip, "decompress_gzip")):
        z = comp(d)
        assert getattr(rsx.Decompressor(), dec)(z, len(d)) == d
    with pytest.raises(ValueError, match="Compression level must be between 0 and 12"):
        rsx.Compressor(13)
    buf = bytearray(10)
    with pytest.raises(RuntimeError, match="Insufficient space"):
        c.compress_deflate_into(d, buf)
    with pytest.raises(RuntimeError, match="Compression failed"):
        c.compress_zlib_into(d, buf)
    dd = rsx.Decompressor()
    with pytest.raises(ValueError, match="exceeds safety limit"):
        dd.decompress_deflate(b"\x03\x00", 10**9)   # tests/security_limit.rs
    dd.set_max_memory_limit(100)
    with pytest.raises(ValueError, match="exceeds maximum memory limit"):
        dd.decompress_deflate(c.compress_deflate(d), len(d))
    with pytest.raises(RuntimeError, match="Decompression failed"):
        rsx.Decompressor().decompress_deflate(b"\x00\x01\x02\x03\x04\x05", 100)
    assert c.deflate_compress_bound(65536) == 65556 and c.gzip_compress_bound(0) == 33


def test_sync_flushed_pieces_concatenate(rsx, eng):
    """rsx_batch_compress_flush (the encoder step of src/stream.rs:42-199): pieces compressed with
    FLUSH_SYNC in separate calls and a last piece with FLUSH_FINISH form one raw DEFLATE stream that
    the oracle inflates to the concatenated input; a sync-flushed piece alone is not a finished stream."""
    pieces = [corpus.text(700000, 41), corpus.mixed(300000, 42), b"", b"x", corpus.text(5000, 43)]
    for level in (1, 6, 12):
        out, st = eng.batch_compress_flush(pieces[:-1], level, rsx.FLUSH_SYNC)
        assert list(st) == [0] * (len(pieces) - 1)
        last, st = eng.batch_compress_flush(pieces[-1:], level, rsx.FLUSH_FINISH)
        assert list(st) == [0]
        stream = b"".join(out) + last[0]
        want = b"".join(pieces)
        assert o.decompress(stream, len(want)) == (o.SUCCESS, len(stream), want)
        # every sync-flushed piece ends on a byte boundary with the empty stored block
        for piece_out in out:
            assert piece_out.endswith(b"\x00\x00\xff\xff")
        r, _, _ = o.decompress(out[0], len(pieces[0]))
        assert r != o.SUCCESS
        # FLUSH_FINISH is the ordinary raw compress
        same, _ = eng.batch_compress(pieces[-1:], rsx.FORMAT_DEFLATE, level)
        assert same[0] == last[0]


@pytest.mark.parametrize("level", [0, 1, 6, 12])
def test_sizes_around_every_internal_boundary(eng, level):
    """Lengths just below, at and just above the engine's internal units -- the 1024-position batch,
    the 64 KiB block, the stored-block limit of 65535, the 256 KiB chunk -- for compressible text,
    zeros, a short period and incompressible bytes; checked by the oracle's and the GPU's inflater."""
    edges = [1, 2, 3, 4, 5, 257, 258, 259, 1023, 1024, 1025, 2047, 2048, 2049, 4095, 4097,
             32767, 32768, 32769, 65534, 65535, 65536, 65537, 131072 + 1, 262143, 262144, 262145,
             262144 + 1024, 2 * 262144 - 1, 2 * 262144 + 1]
    datas = []
    for i, n in enumerate(edges):
        kind = i % 4
        datas.append(corpus.text(n, 900 + i) if kind == 0 else bytes(n) if kind == 1 else
                     corpus.periodic(7, n) if kind == 2 else corpus.random_bytes(n, 950 + i))
    # and the same lengths again with the kinds rotated
    datas += [corpus.text(n, 990 + i) for i, n in enumerate(edges) if i % 4 != 0]
    roundtrip(eng, datas, 0, level)


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_one_large_buffer_in_each_container(eng, fmt):
    """A batch of ONE 64 MiB text buffer (src/api.rs:22-57 with src/compress/mod.rs:699-772: the input is cut
    into 256 chunks of 256 KiB, compressed in parallel and joined by sync-flush markers).  The stream must
    inflate to the input with the oracle, with zlib and with the GPU inflater (one stream, one workgroup)."""
    piece = 8 << 20
    data = b"".join(corpus.text(piece, 500 + k) for k in range(8))
    assert len(data) == 64 << 20
    comp, status = eng.batch_compress([data], fmt, 6)
    assert list(status) == [0]
    c = comp[0]
    assert len(c) <= o.compress_bound(fmt, len(data)) and len(c) < len(data) // 2
    assert c.count(b"\x00\x00\xff\xff") >= 255          # the joints between the 256 chunks
    assert zlib.decompress(c, WBITS[fmt]) == data
    r, used, back = o.decompress(c, len(data), fmt)
    assert (r, used) == (o.SUCCESS, len(c)) and back == data
    back, st = eng.batch_decompress([c], [len(data)], fmt)
    assert list(st) == [0] and back[0] == data


@pytest.mark.parametrize("level", [1, 6, 9, 12])
def test_output_does_not_depend_on_timing(eng, level):
    """The same input gives the same bytes: alone, repeated inside a large batch that fills every CU, and
    in a second call (the match finder's table is filled by atomic max, not in arrival order)."""
    base = [corpus.text(700000, 61), corpus.mixed(600000, 62), corpus.json_like(300000, 63)]
    alone, st = eng.batch_compress(base, 0, level)
    assert list(st) == [0] * len(base)
    crowd = base * 40 if level < 9 else base * 6
    many, st = eng.batch_compress(crowd, 0, level)
    assert list(st) == [0] * len(crowd)
    for i, c in enumerate(many):
        assert c == alone[i % len(base)]
    again, _ = eng.batch_compress(base, 0, level)
    assert again == alone
