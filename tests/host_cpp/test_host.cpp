// GPU test of the C++ host classes (libdeflate-rsx_amd/host/rsx_deflate.hpp).  The cases
// follow the reference's own tests for this surface (tests/batch_test.rs, the api.rs error
// paths exercised by tests/security_limit.rs) and every result is checked against the CPU
// oracle (oracle/rsx_oracle.h): what the GPU compresses must inflate to the input under
// the oracle, what the oracle compresses must decode bit-exactly on the GPU.
//
// usage: test_host            run everything (needs a GPU)
//        test_host --no-gpu   only the argument checks that never reach the engine
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rsx_deflate.hpp"
#include "rsx_oracle.h"

static int g_failed = 0;
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
            g_failed++;                                                              \
        }                                                                            \
    } while (0)

using Buf = std::vector<uint8_t>;

static Buf bytes_of(const char *s) { return Buf(s, s + std::strlen(s)); }

// deterministic data: words from a small vocabulary (compressible) or xorshift bytes
static Buf make_text(size_t n, uint64_t seed)
{
    static const char *words[] = {"the", "batch", "of", "streams", "window", "match", "literal", "huffman",
                                  "block", "offset", "length", "engine", "and", "a", "gpu", "deflate"};
    Buf out;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    while (out.size() < n) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        const char *w = words[(x >> 20) & 15];
        out.insert(out.end(), w, w + std::strlen(w));
        out.push_back((x & 0xF00) ? ' ' : '\n');
    }
    out.resize(n);
    return out;
}
static Buf make_noise(size_t n, uint64_t seed)
{
    Buf out(n);
    uint64_t x = seed * 0xD1342543DE82EF95ull + 7;
    for (size_t i = 0; i < n; i++) {
        x ^= x << 13;
        x ^= x >> 7;
        x ^= x << 17;
        out[i] = (uint8_t)(x >> 32);
    }
    return out;
}

static Buf oracle_compress(const Buf &in, int level, int fmt)
{
    orc_compressor *c = orc_compressor_new(level);
    Buf out(orc_compress_bound(fmt, in.size()));
    size_t n = 0;
    const int r = orc_compress(c, fmt, in.data(), in.size(), out.data(), out.size(), &n);
    orc_compressor_free(c);
    CHECK(r == 0);
    out.resize(n);
    return out;
}
static bool oracle_inflates_to(const Buf &stream, const Buf &want, int fmt)
{
    Buf out(want.size() + 1);
    size_t used = 0, produced = 0;
    const int r = orc_decompress(fmt, stream.data(), stream.size(), out.data(), want.size(), &used, &produced);
    return r == 0 && produced == want.size() && std::memcmp(out.data(), want.data(), want.size()) == 0;
}

static std::vector<rsx::Bytes> refs(const std::vector<Buf> &v)
{
    std::vector<rsx::Bytes> r;
    for (const Buf &b : v) r.emplace_back(b);
    return r;
}

static void test_argument_errors()
{
    // src/api.rs:11-16
    for (int level : {-1, 13, 100}) {
        try {
            rsx::Compressor c(level);
            CHECK(!"level accepted");
        } catch (const rsx::Error &e) {
            CHECK(e.kind() == rsx::ErrorKind::InvalidInput);
            CHECK(std::string(e.what()) == "Compression level must be between 0 and 12");
        }
    }
    // the guards run before anything touches the engine (src/api.rs:213-239)
    rsx::Decompressor d;
    Buf tiny(10);
    try {
        d.decompress_deflate(tiny, 10 * 2000 + 4096 + 1);
        CHECK(!"ratio guard");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::InvalidInput);
        CHECK(std::string(e.what()) == "Expected size 24097 exceeds safety limit for input size 10");
    }
    d.set_max_memory_limit(1000);
    try {
        d.decompress_zlib(tiny, 1001);
        CHECK(!"memory guard");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::InvalidInput);
        CHECK(std::string(e.what()) == "Expected size 1001 exceeds maximum memory limit 1000");
    }
    // overlap check of the _into forms (src/api.rs:114-119, :277-282, :303-315)
    Buf both(64);
    rsx::Compressor c(6);
    try {
        c.compress_deflate_into(rsx::Bytes(both.data(), 32), rsx::BytesMut(both.data() + 16, 48));
        CHECK(!"overlap");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::InvalidInput);
        CHECK(std::string(e.what()) == "Input and output buffers overlap");
    }
    rsx::Decompressor d2;
    try {
        d2.decompress_gzip_into(rsx::Bytes(both.data(), 32), rsx::BytesMut(both.data() + 31, 8));
        CHECK(!"overlap");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::InvalidInput);
    }
    CHECK(c.deflate_compress_bound(1000) == orc_compress_bound(ORC_FMT_DEFLATE, 1000));
    CHECK(c.zlib_compress_bound(0) == orc_compress_bound(ORC_FMT_ZLIB, 0));
    CHECK(c.gzip_compress_bound(123456789) == orc_compress_bound(ORC_FMT_GZIP, 123456789));
}

static void test_batch_roundtrip()
{
    std::vector<Buf> inputs = {bytes_of("Hello world! This is a test string for deflate compression."),
                               bytes_of("Another test string."),
                               bytes_of("Repeating pattern repeating pattern repeating pattern repeating pattern."),
                               bytes_of("Short"),
                               Buf(1000, 0),
                               make_text(300000, 3),
                               make_noise(70000, 4),
                               Buf()};
    for (size_t level : {1, 6, 9, 12}) {
        rsx::BatchCompressor bc(level);
        auto comp = bc.compress_batch(refs(inputs));
        CHECK(comp.size() == inputs.size());
        std::vector<size_t> caps;
        for (size_t i = 0; i < inputs.size(); i++) {
            CHECK(!comp[i].empty());   // even "" has a stream (tests/batch_test.rs:57)
            CHECK(oracle_inflates_to(comp[i], inputs[i], ORC_FMT_DEFLATE));
            caps.push_back(inputs[i].size());
        }
        rsx::BatchDecompressor bd;
        auto back = bd.decompress_batch(refs(comp), caps);
        CHECK(back.size() == inputs.size());
        for (size_t i = 0; i < inputs.size(); i++) CHECK(back[i].has_value() && *back[i] == inputs[i]);
    }
    // empty batches (tests/batch_test.rs:43-52)
    CHECK(rsx::BatchCompressor(6).compress_batch({}).empty());
    CHECK(rsx::BatchDecompressor().decompress_batch({}, {}).empty());
}

static void test_batch_decode_oracle_streams()
{
    std::vector<Buf> texts = {make_text(1 << 20, 11), make_text(65536, 12), make_noise(65536, 13), Buf(40000, 'a'),
                              bytes_of("x")};
    for (int fmt : {ORC_FMT_DEFLATE, ORC_FMT_ZLIB, ORC_FMT_GZIP}) {
        std::vector<Buf> streams;
        std::vector<size_t> caps;
        for (size_t i = 0; i < texts.size(); i++) {
            const Buf &t = texts[i];
            // the reference has no stored-block fallback above level 0 (src/compress/mod.rs:775),
            // so incompressible bytes overflow its bound: those go in as stored blocks
            streams.push_back(oracle_compress(t, i == 2 ? 0 : 6, fmt));
            caps.push_back(t.size() + 100);   // an upper bound, not the exact size
        }
        rsx::BatchDecompressor bd(nullptr, (rsx::Format)fmt);
        auto got = bd.decompress_batch(refs(streams), caps);
        for (size_t i = 0; i < texts.size(); i++) CHECK(got[i].has_value() && *got[i] == texts[i]);
    }
}

static void test_batch_failures()
{
    // garbage and a too-small slot give None for that item only (tests/batch_test.rs:78-110)
    Buf good_text = bytes_of("Hello world!");
    Buf good = oracle_compress(good_text, 6, ORC_FMT_DEFLATE);
    Buf invalid = {0, 1, 2, 3, 4, 5};
    Buf cut(good.begin(), good.begin() + good.size() / 2);
    rsx::BatchDecompressor bd;
    auto got = bd.decompress_batch({rsx::Bytes(invalid), rsx::Bytes(good), rsx::Bytes(good), rsx::Bytes(cut)},
                                   {100, good_text.size() - 1, good_text.size(), 100});
    CHECK(got.size() == 4);
    CHECK(!got[0].has_value());
    CHECK(!got[1].has_value());
    CHECK(got[2].has_value() && *got[2] == good_text);
    CHECK(!got[3].has_value());
}

static void test_one_shot_api()
{
    Buf text = make_text(200000, 21);
    rsx::Compressor c(6);
    rsx::Decompressor d;
    Buf z = c.compress_zlib(text), g = c.compress_gzip(text), r = c.compress_deflate(text);
    CHECK(oracle_inflates_to(z, text, ORC_FMT_ZLIB));
    CHECK(oracle_inflates_to(g, text, ORC_FMT_GZIP));
    CHECK(oracle_inflates_to(r, text, ORC_FMT_DEFLATE));
    CHECK(d.decompress_zlib(z, text.size()) == text);
    CHECK(d.decompress_gzip(g, text.size()) == text);
    CHECK(d.decompress_deflate(r, text.size()) == text);
    // _into forms: exact fit, then one byte short
    Buf out(c.gzip_compress_bound(text.size()));
    const size_t n = c.compress_gzip_into(text, out);
    CHECK(n == g.size() && std::memcmp(out.data(), g.data(), n) == 0);
    Buf small(g.size() - 1);
    try {
        c.compress_gzip_into(text, small);
        CHECK(!"fits?");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::Other && std::string(e.what()) == "Compression failed");
    }
    try {
        c.compress_deflate_into(text, rsx::BytesMut(small.data(), 10));
        CHECK(!"fits?");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::Other && std::string(e.what()) == "Insufficient space");
    }
    Buf back(text.size());
    CHECK(d.decompress_gzip_into(g, back) == text.size() && back == text);
    // a corrupted trailer is InvalidData (src/api.rs:258-261)
    Buf bad = z;
    bad[bad.size() - 1] ^= 1;
    try {
        d.decompress_zlib(bad, text.size());
        CHECK(!"accepted");
    } catch (const rsx::Error &e) {
        CHECK(e.kind() == rsx::ErrorKind::InvalidData && std::string(e.what()) == "Decompression failed");
    }
    // checksums against the oracle, chained in two pieces
    const uint32_t c1 = rsx::crc32(0, rsx::Bytes(text.data(), 1000));
    CHECK(rsx::crc32(c1, rsx::Bytes(text.data() + 1000, text.size() - 1000)) == orc_crc32(0, text.data(), text.size()));
    const uint32_t a1 = rsx::adler32(1, rsx::Bytes(text.data(), 77777));
    CHECK(rsx::adler32(a1, rsx::Bytes(text.data() + 77777, text.size() - 77777)) ==
          orc_adler32(1, text.data(), text.size()));
    CHECK(rsx::crc32(0, Buf()) == 0 && rsx::adler32(1, Buf()) == 1);
}

// src/stream.rs: what the encoder writes is one raw DEFLATE stream, whatever the write and
// flush pattern; the oracle must inflate it to everything that was written
static void test_stream_encoder()
{
    Buf text = make_text(3 * 1024 * 1024 + 12345, 31);
    Buf noise = make_noise(300000, 32);
    for (size_t level : {1, 6}) {
        Buf sink;
        auto writer = [&sink](const uint8_t *p, size_t n) { sink.insert(sink.end(), p, p + n); };
        Buf all;
        {
            rsx::DeflateEncoder<decltype(writer)> enc(writer, level);
            size_t at = 0, step = 1;
            while (at < text.size()) {   // ragged writes, from 1 byte to ~700 KB
                const size_t n = std::min(step, text.size() - at);
                CHECK(enc.write(rsx::Bytes(text.data() + at, n)) == n);
                all.insert(all.end(), text.begin() + at, text.begin() + at + n);
                at += n;
                step = step * 7 + 3;
                if (step > 700000) step = 11;
            }
            enc.flush();
            const size_t after_flush = sink.size();
            enc.flush();                       // nothing buffered: writes nothing (src/stream.rs:43-45)
            CHECK(sink.size() == after_flush);
            enc.write(noise);                  // incompressible piece in the middle of the stream
            all.insert(all.end(), noise.begin(), noise.end());
            enc.write(bytes_of("tail"));
            const Buf tail = bytes_of("tail");
            all.insert(all.end(), tail.begin(), tail.end());
            enc.finish();
        }
        CHECK(oracle_inflates_to(sink, all, ORC_FMT_DEFLATE));
        // and the GPU decodes its own stream
        rsx::Decompressor d;
        CHECK(d.decompress_deflate(sink, all.size()) == all);
    }
    // an encoder that is dropped still ends its stream; an empty one is a valid empty stream
    Buf sink;
    auto writer = [&sink](const uint8_t *p, size_t n) { sink.insert(sink.end(), p, p + n); };
    {
        rsx::DeflateEncoder<decltype(writer)> enc(writer, 6);
        enc.with_buffer_size(1000).write(make_text(2500, 5));
    }
    CHECK(oracle_inflates_to(sink, make_text(2500, 5), ORC_FMT_DEFLATE));
    sink.clear();
    {
        rsx::DeflateEncoder<decltype(writer)> enc(writer, 6);
        enc.finish();
    }
    CHECK(!sink.empty() && oracle_inflates_to(sink, Buf(), ORC_FMT_DEFLATE));
    // a slot of exactly the bound holds incompressible input of several chunks
    Buf big = make_noise(1 << 20, 33);
    rsx::Compressor c(6);
    Buf z = c.compress_deflate(big);
    CHECK(oracle_inflates_to(z, big, ORC_FMT_DEFLATE));
}

// src/stream.rs:235-377: the decoder hands out, through reads of any size, exactly the bytes the stream
// holds; its errors carry the reference's kinds and messages
static void test_stream_decoder()
{
    Buf text = make_text(2 * 1024 * 1024 + 777, 41);
    rsx::Compressor c(6);
    const Buf z = c.compress_deflate(text);   // (several chunks: the decoder cuts the stream at their joints)
    auto reader_over = [](const Buf &src, size_t step) {
        auto at = std::make_shared<size_t>(0);
        return [&src, at, step](uint8_t *dst, size_t cap) {
            const size_t n = std::min(std::min(cap, step), src.size() - *at);
            if (n) std::memcpy(dst, src.data() + *at, n);
            *at += n;
            return n;
        };
    };
    for (size_t step : {size_t(1) << 30, size_t(4097), size_t(100000)}) {
        auto reader = reader_over(z, step);
        rsx::DeflateDecoder<decltype(reader)> dec(reader);
        Buf got, piece(70001);
        for (;;) {   // ragged reads
            const size_t n = dec.read(rsx::BytesMut(piece.data(), 1 + got.size() % piece.size()));
            if (n == 0) break;
            got.insert(got.end(), piece.begin(), piece.begin() + n);
        }
        CHECK(got == text);
        CHECK(dec.read(piece) == 0);   // stays at the end
    }
    {
        auto reader = reader_over(z, 65536);
        rsx::DeflateDecoder<decltype(reader)> dec(reader);
        Buf head(1000);
        CHECK(dec.read(head) == 1000 && std::equal(head.begin(), head.end(), text.begin()));
        const Buf rest = dec.read_to_end();
        CHECK(rest.size() == text.size() - 1000 && std::equal(rest.begin(), rest.end(), text.begin() + 1000));
    }
    const Buf empty;
    {
        auto reader = reader_over(empty, 100);
        rsx::DeflateDecoder<decltype(reader)> dec(reader);
        Buf piece(10);
        CHECK(dec.read(piece) == 0);   // an empty reader is an empty stream (src/stream.rs:365-371)
    }
    const Buf cut(z.begin(), z.begin() + z.size() / 2);
    Buf damaged = z;
    damaged[damaged.size() / 3] ^= 0x40;
    damaged[damaged.size() / 3 + 1] ^= 0x04;
    for (int which = 0; which < 2; which++) {
        const Buf &src = which ? damaged : cut;
        auto reader = reader_over(src, 50000);
        rsx::DeflateDecoder<decltype(reader)> dec(reader);
        // what the oracle says about the stream decides which error it must be; the bytes in front of the
        // error are the text's (src/stream.rs:316-333: what was decoded goes out before the error does)
        std::vector<uint8_t> scratch(text.size());
        size_t used = 0, produced = 0;
        const int r = orc_decompress(ORC_FMT_DEFLATE, src.data(), src.size(), scratch.data(), scratch.size(), &used, &produced);
        Buf got, piece(30000);
        bool threw = false;
        try {
            for (;;) {
                const size_t n = dec.read(piece);
                if (n == 0) break;
                got.insert(got.end(), piece.begin(), piece.begin() + n);
            }
        } catch (const rsx::Error &e) {
            threw = true;
            if (r == 4) CHECK(e.kind() == rsx::ErrorKind::UnexpectedEof && std::string(e.what()) == "unexpected EOF");
            else CHECK(r != 0 && e.kind() == rsx::ErrorKind::InvalidData && std::string(e.what()) == "deflate decompression failed");
        }
        CHECK(threw == (r != 0));
        if (r == 0) {   // (damage that still decodes: the oracle's bytes)
            CHECK(got.size() == produced && std::equal(got.begin(), got.end(), scratch.begin()));
            continue;
        }
        // whole 64 KiB blocks in front of the cut / the damage went out first, and they are the text's
        const size_t sure = text.size() / 4;
        CHECK(got.size() >= sure && std::equal(got.begin(), got.begin() + (std::ptrdiff_t)sure, text.begin()));
        try {   // the error stays
            dec.read(piece);
            CHECK(false);
        } catch (const rsx::Error &) {
        }
    }
    {
        // bounded memory: a stream of 24 MiB goes through buffers of a few MiB
        Buf big;
        for (int k = 0; k < 12; k++) {
            const Buf t = make_text(2 * 1024 * 1024, 50 + k);
            big.insert(big.end(), t.begin(), t.end());
        }
        const Buf zb = c.compress_deflate(big);
        auto reader = reader_over(zb, 300000);
        rsx::DeflateDecoder<decltype(reader)> dec(reader);
        Buf piece(1 << 20);
        size_t at = 0, most = 0;
        bool same = true;
        for (;;) {
            const size_t n = dec.read(piece);
            if (n == 0) break;
            same = same && at + n <= big.size() && std::equal(piece.begin(), piece.begin() + n, big.begin() + at);
            at += n;
            most = std::max(most, dec.buffered_bytes());
        }
        CHECK(same && at == big.size());
        CHECK(most < (12u << 20));
    }
}

int main(int argc, char **argv)
{
    const bool no_gpu = argc > 1 && std::strcmp(argv[1], "--no-gpu") == 0;
    test_argument_errors();
    if (!no_gpu) {
        try {
            test_batch_roundtrip();
            test_batch_decode_oracle_streams();
            test_batch_failures();
            test_one_shot_api();
            test_stream_encoder();
            test_stream_decoder();
        } catch (const std::exception &e) {
            std::fprintf(stderr, "FAILED: unexpected exception: %s\n", e.what());
            g_failed++;
        }
    }
    if (g_failed) {
        std::fprintf(stderr, "%d check(s) failed\n", g_failed);
        return 1;
    }
    std::printf("host c++ tests ok%s\n", no_gpu ? " (argument checks only)" : "");
    return 0;
}
