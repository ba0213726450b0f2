// rsx_deflate.hpp -- C++17 host side of the MI355X batch DEFLATE engine.
//
// Header-only mirror of the reference crate's public surface for the batch path, written
// over the C ABI of librsx_deflate.so (include/rsx_deflate.h):
//
//   rsx::BatchCompressor / rsx::BatchDecompressor   src/batch.rs:5-58, :60-101
//   rsx::Compressor / rsx::Decompressor             src/api.rs:5-133, :135-301
//   rsx::DeflateEncoder                             src/stream.rs:15-233
//   rsx::crc32 / rsx::adler32                       src/lib.rs:15-17
//
// Same names, argument meaning and error behaviour.  io::Error becomes rsx::Error with the
// reference's ErrorKind and message; Vec<u8> becomes std::vector<uint8_t>; Option<Vec<u8>>
// becomes std::optional.  There is no CPU path behind these classes: if the engine cannot
// be created or a launch fails, the call throws rsx::EngineError (the reference would fall
// back to its rayon loop, src/batch.rs:28-30; a caller that wants that keeps its own).
//
// Link with -lrsx_deflate.  One Engine per GPU; calls on one engine serialise inside the
// library, so the classes may be shared between threads like their `&self` originals.
#ifndef RSX_DEFLATE_HPP
#define RSX_DEFLATE_HPP

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rsx_deflate.h"

namespace rsx {

// a borrowed byte range (&[u8])
struct Bytes {
    const uint8_t *data = nullptr;
    size_t size = 0;
    Bytes() = default;
    Bytes(const uint8_t *d, size_t n) : data(d), size(n) {}
    Bytes(const std::vector<uint8_t> &v) : data(v.data()), size(v.size()) {}
    Bytes(const std::string &s) : data(reinterpret_cast<const uint8_t *>(s.data())), size(s.size()) {}
};
// a borrowed writable byte range (&mut [u8])
struct BytesMut {
    uint8_t *data = nullptr;
    size_t size = 0;
    BytesMut() = default;
    BytesMut(uint8_t *d, size_t n) : data(d), size(n) {}
    BytesMut(std::vector<uint8_t> &v) : data(v.data()), size(v.size()) {}
};

// io::ErrorKind values the reference's api.rs produces
enum class ErrorKind { InvalidInput, InvalidData, UnexpectedEof, Other };

class Error : public std::runtime_error {
  public:
    Error(ErrorKind k, const std::string &msg) : std::runtime_error(msg), kind_(k) {}
    ErrorKind kind() const { return kind_; }

  private:
    ErrorKind kind_;
};

// the accelerator itself failed (no GPU, HIP error, out of device memory)
class EngineError : public std::runtime_error {
  public:
    EngineError(int code, const std::string &msg) : std::runtime_error(msg), code_(code) {}
    int code() const { return code_; }

  private:
    int code_;
};

enum class Format : int { Deflate = RSX_FORMAT_DEFLATE, Zlib = RSX_FORMAT_ZLIB, Gzip = RSX_FORMAT_GZIP };

// Owns one rsx_engine (one GPU).  Takes the place of CudaBatchCompressor::new
// (src/batch_cuda.rs:20-41).
class Engine {
  public:
    explicit Engine(int device = 0, void *hip_stream = nullptr)
    {
        const int r = rsx_engine_create(device, hip_stream, &h_);
        if (r != RSX_OK || !h_) throw EngineError(r, "rsx_engine_create failed (no usable GPU?)");
    }
    ~Engine() { rsx_engine_destroy(h_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;

    rsx_engine *handle() const { return h_; }
    void check(int r) const
    {
        if (r != RSX_OK) throw EngineError(r, std::string("rsx engine: ") + rsx_engine_last_error(h_));
    }

    // the process-wide engine on GPU 0, created on first use (the reference's OnceLock,
    // src/batch.rs:7-8)
    static std::shared_ptr<Engine> shared()
    {
        static std::shared_ptr<Engine> e = std::make_shared<Engine>(0);
        return e;
    }

  private:
    rsx_engine *h_ = nullptr;
};

namespace detail {

inline bool is_overlapping(const uint8_t *a, size_t an, const uint8_t *b, size_t bn)  // src/api.rs:303-315
{
    const uintptr_t p1 = reinterpret_cast<uintptr_t>(a), p2 = reinterpret_cast<uintptr_t>(b);
    const uintptr_t max = std::numeric_limits<uintptr_t>::max();
    const uintptr_t e1 = an > max - p1 ? max : p1 + an, e2 = bn > max - p2 ? max : p2 + bn;
    return (p1 > p2 ? p1 : p2) < (e1 < e2 ? e1 : e2);
}

// flat copy of a batch + its n+1 offsets: the layout src/batch_cuda.rs:57-87 builds
struct Flat {
    std::vector<uint8_t> bytes;
    std::vector<uint64_t> off;
    explicit Flat(const std::vector<Bytes> &items)
    {
        off.resize(items.size() + 1);
        uint64_t total = 0;
        for (size_t i = 0; i < items.size(); i++) {
            off[i] = total;
            total += items[i].size;
        }
        off[items.size()] = total;
        bytes.resize(total ? total : 1);
        for (size_t i = 0; i < items.size(); i++)
            if (items[i].size) std::copy(items[i].data, items[i].data + items[i].size, bytes.begin() + off[i]);
    }
};

inline std::vector<uint64_t> slots(const std::vector<uint64_t> &caps)
{
    std::vector<uint64_t> off(caps.size() + 1);
    uint64_t total = 0;
    for (size_t i = 0; i < caps.size(); i++) {
        off[i] = total;
        total += caps[i];
    }
    off[caps.size()] = total;
    return off;
}

}  // namespace detail

// src/batch.rs:5-58.  Raw DEFLATE like the reference; the format argument also gives the
// zlib / gzip framing of src/compress/mod.rs:2248-2357 per item.
class BatchCompressor {
  public:
    explicit BatchCompressor(size_t level, std::shared_ptr<Engine> engine = nullptr,
                             Format format = Format::Deflate)
        : level_(level), format_(format), engine_(std::move(engine))
    {
    }

    // one output per input; an item that fails to compress yields an empty vector
    // (src/batch.rs:52-54)
    std::vector<std::vector<uint8_t>> compress_batch(const std::vector<Bytes> &inputs) const
    {
        const size_t n = inputs.size();
        std::vector<std::vector<uint8_t>> result(n);
        if (n == 0) return result;
        Engine &eng = engine();
        detail::Flat in(inputs);
        std::vector<uint64_t> caps(n);
        for (size_t i = 0; i < n; i++) caps[i] = rsx_compress_bound((int)format_, inputs[i].size);
        const std::vector<uint64_t> out_off = detail::slots(caps);
        std::vector<uint8_t> out(out_off[n] ? out_off[n] : 1);
        std::vector<uint64_t> sizes(n);
        std::vector<int32_t> status(n);
        eng.check(rsx_batch_compress(eng.handle(), (int)format_, (int)level_, in.bytes.data(), in.off.data(), n,
                                     out.data(), out_off.data(), sizes.data(), status.data()));
        for (size_t i = 0; i < n; i++) {
            // sizes come from the device: never trust them past the slot (src/batch_cuda.rs:123-137)
            if (status[i] == RSX_SUCCESS && sizes[i] <= caps[i])
                result[i].assign(out.begin() + out_off[i], out.begin() + out_off[i] + sizes[i]);
        }
        return result;
    }

  private:
    Engine &engine() const
    {
        if (!engine_) engine_ = Engine::shared();
        return *engine_;
    }
    size_t level_;
    Format format_;
    mutable std::shared_ptr<Engine> engine_;
};

// src/batch.rs:60-101
class BatchDecompressor {
  public:
    explicit BatchDecompressor(std::shared_ptr<Engine> engine = nullptr, Format format = Format::Deflate)
        : format_(format), engine_(std::move(engine))
    {
    }

    // max_out_sizes[i] is an upper bound for item i; the result holds exactly the bytes
    // produced, or nullopt when the item does not decode (src/batch.rs:94-97).  Like the
    // reference's zip, surplus entries of the longer argument are ignored.
    std::vector<std::optional<std::vector<uint8_t>>> decompress_batch(const std::vector<Bytes> &inputs,
                                                                      const std::vector<size_t> &max_out_sizes) const
    {
        const size_t n = inputs.size() < max_out_sizes.size() ? inputs.size() : max_out_sizes.size();
        std::vector<std::optional<std::vector<uint8_t>>> result(n);
        if (n == 0) return result;
        Engine &eng = engine();
        detail::Flat in(std::vector<Bytes>(inputs.begin(), inputs.begin() + n));
        std::vector<uint64_t> caps(max_out_sizes.begin(), max_out_sizes.begin() + n);
        const std::vector<uint64_t> out_off = detail::slots(caps);
        std::vector<uint8_t> out(out_off[n] ? out_off[n] : 1);
        std::vector<uint64_t> sizes(n);
        std::vector<int32_t> status(n);
        eng.check(rsx_batch_decompress(eng.handle(), (int)format_, in.bytes.data(), in.off.data(), n, out.data(),
                                       out_off.data(), sizes.data(), status.data()));
        for (size_t i = 0; i < n; i++) {
            if (status[i] == RSX_SUCCESS && sizes[i] <= caps[i])
                result[i].emplace(out.begin() + out_off[i], out.begin() + out_off[i] + sizes[i]);
        }
        return result;
    }

  private:
    Engine &engine() const
    {
        if (!engine_) engine_ = Engine::shared();
        return *engine_;
    }
    Format format_;
    mutable std::shared_ptr<Engine> engine_;
};

// src/api.rs:5-133
class Compressor {
  public:
    explicit Compressor(int level, std::shared_ptr<Engine> engine = nullptr) : level_(level), engine_(std::move(engine))
    {
        if (level < 0 || level > 12)  // src/api.rs:11-16
            throw Error(ErrorKind::InvalidInput, "Compression level must be between 0 and 12");
    }

    std::vector<uint8_t> compress_deflate(Bytes data) { return to_vec(data, Format::Deflate); }
    std::vector<uint8_t> compress_zlib(Bytes data) { return to_vec(data, Format::Zlib); }
    std::vector<uint8_t> compress_gzip(Bytes data) { return to_vec(data, Format::Gzip); }

    size_t compress_deflate_into(Bytes data, BytesMut output)
    {
        return into(data, output, Format::Deflate, "Insufficient space");  // src/api.rs:31
    }
    size_t compress_zlib_into(Bytes data, BytesMut output)
    {
        return into(data, output, Format::Zlib, "Compression failed");  // src/api.rs:43
    }
    size_t compress_gzip_into(Bytes data, BytesMut output)
    {
        return into(data, output, Format::Gzip, "Compression failed");  // src/api.rs:54
    }

    size_t deflate_compress_bound(size_t size) { return rsx_compress_bound(RSX_FORMAT_DEFLATE, size); }
    size_t zlib_compress_bound(size_t size) { return rsx_compress_bound(RSX_FORMAT_ZLIB, size); }
    size_t gzip_compress_bound(size_t size) { return rsx_compress_bound(RSX_FORMAT_GZIP, size); }

  private:
    // one item through the batch entry point; returns false for InsufficientSpace
    bool run(Bytes data, Format f, uint8_t *out, uint64_t cap, uint64_t *size)
    {
        if (!engine_) engine_ = Engine::shared();
        const uint64_t in_off[2] = {0, data.size}, out_off[2] = {0, cap};
        const uint8_t dummy = 0;
        uint8_t sink = 0;
        int32_t status = 0;
        engine_->check(rsx_batch_compress(engine_->handle(), (int)f, level_, data.size ? data.data : &dummy, in_off, 1,
                                          cap ? out : &sink, out_off, size, &status));
        return status == RSX_SUCCESS && *size <= cap;
    }
    std::vector<uint8_t> to_vec(Bytes data, Format f)  // compress_helper, src/api.rs:71-98
    {
        std::vector<uint8_t> out(rsx_compress_bound((int)f, data.size));
        uint64_t size = 0;
        if (!run(data, f, out.data(), out.size(), &size)) throw Error(ErrorKind::Other, "Insufficient space");
        out.resize(size);
        return out;
    }
    size_t into(Bytes data, BytesMut output, Format f, const char *msg)  // compress_into_helper, :100-131
    {
        if (detail::is_overlapping(data.data, data.size, output.data, output.size))
            throw Error(ErrorKind::InvalidInput, "Input and output buffers overlap");
        uint64_t size = 0;
        if (!run(data, f, output.data, output.size, &size)) throw Error(ErrorKind::Other, msg);
        return size;
    }
    int level_;
    std::shared_ptr<Engine> engine_;
};

// src/api.rs:135-301: the expected-size API with its zip-bomb guards (:213-239)
class Decompressor {
  public:
    explicit Decompressor(std::shared_ptr<Engine> engine = nullptr) : engine_(std::move(engine)) {}

    void set_max_memory_limit(size_t limit) { max_memory_limit_ = limit; }
    void set_limit_ratio(size_t ratio) { limit_ratio_ = ratio; }

    std::vector<uint8_t> decompress_deflate(Bytes data, size_t expected_size)
    {
        return guarded(data, expected_size, Format::Deflate);
    }
    std::vector<uint8_t> decompress_zlib(Bytes data, size_t expected_size)
    {
        return guarded(data, expected_size, Format::Zlib);
    }
    std::vector<uint8_t> decompress_gzip(Bytes data, size_t expected_size)
    {
        return guarded(data, expected_size, Format::Gzip);
    }
    // the _into forms skip the guards (tests/security_limit.rs:274-297)
    size_t decompress_deflate_into(Bytes data, BytesMut output) { return into(data, output, Format::Deflate); }
    size_t decompress_zlib_into(Bytes data, BytesMut output) { return into(data, output, Format::Zlib); }
    size_t decompress_gzip_into(Bytes data, BytesMut output) { return into(data, output, Format::Gzip); }

  private:
    bool run(Bytes data, Format f, uint8_t *out, uint64_t cap, uint64_t *size)
    {
        if (!engine_) engine_ = Engine::shared();
        const uint64_t in_off[2] = {0, data.size}, out_off[2] = {0, cap};
        const uint8_t dummy = 0;
        uint8_t sink = 0;
        int32_t status = 0;
        engine_->check(rsx_batch_decompress(engine_->handle(), (int)f, data.size ? data.data : &dummy, in_off, 1,
                                            cap ? out : &sink, out_off, size, &status));
        return status == RSX_SUCCESS && *size <= cap;
    }
    std::vector<uint8_t> guarded(Bytes data, size_t expected_size, Format f)  // decompress_helper, :200-263
    {
        const size_t max = std::numeric_limits<size_t>::max();
        size_t limit = limit_ratio_ != 0 && data.size > max / limit_ratio_ ? max : data.size * limit_ratio_;
        limit = limit > max - 4096 ? max : limit + 4096;
        if (expected_size > limit)
            throw Error(ErrorKind::InvalidInput, "Expected size " + std::to_string(expected_size) +
                                                     " exceeds safety limit for input size " +
                                                     std::to_string(data.size));
        if (expected_size > max_memory_limit_)
            throw Error(ErrorKind::InvalidInput, "Expected size " + std::to_string(expected_size) +
                                                     " exceeds maximum memory limit " +
                                                     std::to_string(max_memory_limit_));
        std::vector<uint8_t> out(expected_size);
        uint64_t size = 0;
        if (!run(data, f, out.data(), out.size(), &size)) throw Error(ErrorKind::InvalidData, "Decompression failed");
        out.resize(size);
        return out;
    }
    size_t into(Bytes data, BytesMut output, Format f)  // decompress_into_helper, :265-300
    {
        if (detail::is_overlapping(data.data, data.size, output.data, output.size))
            throw Error(ErrorKind::InvalidInput, "Input and output buffers overlap");
        uint64_t size = 0;
        if (!run(data, f, output.data, output.size, &size)) throw Error(ErrorKind::InvalidData, "Decompression failed");
        return size;
    }
    size_t max_memory_limit_ = std::numeric_limits<size_t>::max();
    size_t limit_ratio_ = 2000;
    std::shared_ptr<Engine> engine_;
};

// src/stream.rs:15-233: buffers what is written and compresses the buffer in 256 KiB
// chunks whenever it reaches buffer_size (1 MiB unless changed), on flush() and on
// finish().  All pieces but the last end with a sync flush, so what reaches the writer is
// one raw DEFLATE stream.  Writer is anything callable as writer(const uint8_t *, size_t).
// finish() must be called to end the stream; the destructor tries and swallows errors,
// like the reference's Drop (src/stream.rs:227-233).
template <class Writer>
class DeflateEncoder {
  public:
    DeflateEncoder(Writer writer, size_t level, std::shared_ptr<Engine> engine = nullptr)
        : writer_(std::move(writer)), level_(level), engine_(std::move(engine))
    {
        buffer_.reserve(buffer_size_);
    }
    DeflateEncoder(const DeflateEncoder &) = delete;
    DeflateEncoder &operator=(const DeflateEncoder &) = delete;
    ~DeflateEncoder()
    {
        if (!finished_) {
            try {
                flush_buffer(true);
            } catch (...) {
            }
        }
    }

    DeflateEncoder &with_buffer_size(size_t size)
    {
        buffer_size_ = size;
        buffer_.reserve(size);
        return *this;
    }
    // io::Write::write: takes everything, compresses when the buffer is full
    size_t write(Bytes buf)
    {
        buffer_.insert(buffer_.end(), buf.data, buf.data + buf.size);
        if (buffer_.size() >= buffer_size_) flush_buffer(false);
        return buf.size;
    }
    // io::Write::flush: everything written so far becomes decodable
    void flush() { flush_buffer(false); }
    // ends the stream and hands the writer back
    Writer finish()
    {
        flush_buffer(true);
        return std::move(writer_);
    }

  private:
    void flush_buffer(bool final_block)  // src/stream.rs:42-199
    {
        if (finished_) return;
        if (buffer_.empty() && !final_block) return;
        if (!engine_) engine_ = Engine::shared();
        const uint64_t cap = rsx_compress_bound(RSX_FORMAT_DEFLATE, buffer_.size()) + 5;
        out_.resize(cap);
        const uint64_t in_off[2] = {0, buffer_.size()}, out_off[2] = {0, cap};
        const uint8_t dummy = 0;
        uint64_t size = 0;
        int32_t status = 0;
        engine_->check(rsx_batch_compress_flush(engine_->handle(), (int)level_,
                                                final_block ? RSX_FLUSH_FINISH : RSX_FLUSH_SYNC,
                                                buffer_.empty() ? &dummy : buffer_.data(), in_off, 1, out_.data(),
                                                out_off, &size, &status));
        if (status != RSX_SUCCESS || size > cap) throw Error(ErrorKind::Other, "Compression failed");
        writer_(out_.data(), (size_t)size);
        buffer_.clear();
        if (final_block) finished_ = true;
    }
    Writer writer_;
    size_t level_;
    std::shared_ptr<Engine> engine_;
    std::vector<uint8_t> buffer_, out_;
    size_t buffer_size_ = 1024 * 1024;
    bool finished_ = false;
};

// libdeflate::stream::DeflateDecoder<R> (src/stream.rs:235-377): reads one raw DEFLATE stream from a
// reader and hands out the decompressed bytes through read().  Reader is anything callable as
// size_t reader(uint8_t *dst, size_t cap) that returns 0 at the end of its data.  Like the reference it
// decodes as it reads, in bounded memory: a buffer of input (1 MiB to begin with), the last 32 KiB of
// output as history, and a buffer of output (4 MiB to begin with).  The reference feeds a CPU state machine
// that can stop at any bit (src/decompress/mod.rs:213-353); the engine's step (rsx_inflate_stream_step)
// stops between two blocks, so a block must fit the buffers whole: they double when a step makes no
// progress (the reference's input buffer doubles up to 1 MiB and then fails with "input buffer full",
// :346-352; here up to 256 MiB).  Bytes and errors are the reference's: BadData -> InvalidData "deflate
// decompression failed" (:329-333) behind the bytes of the blocks in front of the damage, a stream that
// ends early -> UnexpectedEof "unexpected EOF" (:359-370) behind everything that could be decoded, an
// empty reader -> read() returns 0 (:371).
template <class Reader>
class DeflateDecoder {
  public:
    explicit DeflateDecoder(Reader inner, std::shared_ptr<Engine> engine = nullptr)
        : inner_(std::move(inner)), engine_(std::move(engine))
    {
    }
    // io::Read::read: up to buf.size bytes, 0 at the end of the stream
    size_t read(BytesMut buf)
    {
        if (buf.size == 0) return 0;
        while (out_pos_ == out_len_ && !done_) step();
        const size_t count = std::min(buf.size, out_len_ - out_pos_);
        if (count) std::memcpy(buf.data, out_.data() + out_pos_, count);
        out_pos_ += count;
        return count;
    }
    // everything that is left (io::Read::read_to_end)
    std::vector<uint8_t> read_to_end()
    {
        std::vector<uint8_t> rest;
        for (;;) {
            while (out_pos_ == out_len_ && !done_) step();
            if (out_pos_ == out_len_) return rest;
            rest.insert(rest.end(), out_.begin() + (std::ptrdiff_t)out_pos_, out_.begin() + (std::ptrdiff_t)out_len_);
            out_pos_ = out_len_;
        }
    }
    // bytes of buffer memory the decoder holds at the moment (input, history, output)
    size_t buffered_bytes() const { return in_.capacity() + hist_.capacity() + out_.capacity(); }

  private:
    static constexpr size_t kMaxBuffer = size_t(256) << 20;
    // one step of the engine over what has been read so far; called with the output buffer drained
    void step()
    {
        if (failed_) throw Error(fail_kind_, fail_msg_);
        // top the input up to feed_ bytes
        if (in_pos_ && (in_pos_ == in_.size() || in_pos_ >= feed_ / 2)) {
            in_.erase(in_.begin(), in_.begin() + (std::ptrdiff_t)in_pos_);
            in_pos_ = 0;
        }
        while (!eof_ && in_.size() - in_pos_ < feed_) {
            const size_t at = in_.size(), room = feed_ - (at - in_pos_);
            in_.resize(at + room);
            const size_t n = inner_(in_.data() + at, room);
            in_.resize(at + n);
            if (n == 0) eof_ = true;
        }
        if (in_.size() == in_pos_) {
            // no input left: an empty reader is an empty stream, anything else ends early (src/stream.rs:355-371)
            if (!started_) {
                done_ = true;
                return;
            }
            fail(ErrorKind::UnexpectedEof, "unexpected EOF");
        }
        if (!engine_) engine_ = Engine::shared();
        if (out_.size() < out_cap_) out_.resize(out_cap_);
        uint64_t next_bit = 0;
        size_t produced = 0;
        int32_t status = 0;
        engine_->check(rsx_inflate_stream_step(engine_->handle(), in_.data() + in_pos_, in_.size() - in_pos_, bit_,
                                               hist_.data(), hist_.size(), out_.data(), out_cap_, &next_bit, &produced,
                                               &status));
        const bool progress = produced != 0 || next_bit != bit_;
        started_ = started_ || progress || status == RSX_SUCCESS;
        in_pos_ += (size_t)(next_bit >> 3);
        bit_ = (uint32_t)(next_bit & 7);
        out_pos_ = 0;
        out_len_ = produced;
        // the history of the next step: the last 32 KiB of everything produced so far
        if (produced >= 32768) {
            hist_.assign(out_.begin() + (std::ptrdiff_t)(produced - 32768), out_.begin() + (std::ptrdiff_t)produced);
        } else if (produced) {
            const size_t keep = std::min(hist_.size(), 32768 - produced);
            hist_.erase(hist_.begin(), hist_.end() - (std::ptrdiff_t)keep);
            hist_.insert(hist_.end(), out_.begin(), out_.begin() + (std::ptrdiff_t)produced);
        }
        switch (status) {
        case RSX_SUCCESS:
            done_ = true;
            return;
        case RSX_SHORT_INPUT:
            if (progress) return;
            if (eof_) fail(ErrorKind::UnexpectedEof, "unexpected EOF");
            if (feed_ >= kMaxBuffer) fail(ErrorKind::Other, "input buffer full");
            feed_ *= 2;   // the block in front is longer than the buffer
            return;
        case RSX_INSUFFICIENT_SPACE:
            if (progress) return;
            if (out_cap_ >= kMaxBuffer) fail(ErrorKind::Other, "output buffer full");
            out_cap_ *= 2;
            return;
        default:
            // damaged data: the bytes of the blocks in front go out first (src/stream.rs:316-333)
            failed_ = true;
            fail_kind_ = ErrorKind::InvalidData;
            fail_msg_ = "deflate decompression failed";
            if (!produced) throw Error(fail_kind_, fail_msg_);
            return;
        }
    }
    [[noreturn]] void fail(ErrorKind kind, const char *msg)
    {
        failed_ = true;
        fail_kind_ = kind;
        fail_msg_ = msg;
        throw Error(kind, msg);
    }
    Reader inner_;
    std::shared_ptr<Engine> engine_;
    std::vector<uint8_t> in_, hist_, out_;
    size_t in_pos_ = 0, out_pos_ = 0, out_len_ = 0;
    size_t feed_ = size_t(1) << 20, out_cap_ = size_t(4) << 20;
    uint32_t bit_ = 0;
    bool eof_ = false, done_ = false, started_ = false, failed_ = false;
    ErrorKind fail_kind_ = ErrorKind::Other;
    const char *fail_msg_ = "";
};

// crate-level checksums (src/lib.rs:15-17): pass the previous result to continue one
inline uint32_t crc32(uint32_t crc, Bytes data, Engine *engine = nullptr)
{
    std::shared_ptr<Engine> keep = engine ? nullptr : Engine::shared();
    Engine &e = engine ? *engine : *keep;
    const uint8_t dummy = 0;
    uint32_t v = 0;
    e.check(rsx_crc32(e.handle(), crc, data.size ? data.data : &dummy, data.size, &v));
    return v;
}
inline uint32_t adler32(uint32_t adler, Bytes data, Engine *engine = nullptr)
{
    std::shared_ptr<Engine> keep = engine ? nullptr : Engine::shared();
    Engine &e = engine ? *engine : *keep;
    const uint8_t dummy = 0;
    uint32_t v = 0;
    e.check(rsx_adler32(e.handle(), adler, data.size ? data.data : &dummy, data.size, &v));
    return v;
}

}  // namespace rsx
#endif
