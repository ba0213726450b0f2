// api.cpp -- the extern "C" surface declared in include/rsx_deflate.h: engine
// lifetime, workspace management, and the host-pointer wrappers that stage
// through device memory the way src/batch_cuda.rs:81-141 does (H2D, launch,
// D2H) around the *_device entry points.
#include "engine.h"

#include <cstdlib>
#include <cstring>
#include <new>

int ws_reserve(rsx_engine *e, size_t bytes)
{
    e->ws_used = 0;
    e->d_last_parts = nullptr;   // (points into the workspace this call carves anew)
    e->d_last_fallbacks = nullptr;
    e->arena = e->ws;
    e->arena_size = e->ws_size;
    if (bytes <= e->ws_size) return RSX_OK;
    e->arena = nullptr;
    e->arena_size = 0;
    if (e->ws) {
        RSX_HIP(e, hipStreamSynchronize(e->stream));
        RSX_HIP(e, hipFree(e->ws));
        e->ws = nullptr;
        e->ws_size = 0;
    }
    size_t want = align_up(bytes + bytes / 4, 1 << 20);
    if (hipMalloc(reinterpret_cast<void **>(&e->ws), want) != hipSuccess) {
        (void)hipGetLastError();
        e->err = "device workspace allocation of " + std::to_string(want) + " bytes failed";
        return RSX_E_NOMEM;
    }
    e->ws_size = want;
    e->arena = e->ws;
    e->arena_size = want;
    return RSX_OK;
}

int ws_use_static(rsx_engine *e, size_t bytes)
{
    e->ws_used = 0;
    e->d_last_parts = nullptr;
    e->d_last_fallbacks = nullptr;
    e->arena = e->static_scratch;
    e->arena_size = e->static_scratch_size;
    if (!e->static_scratch || bytes > e->static_scratch_size) {
        e->arena = nullptr;
        e->arena_size = 0;
        return RSX_E_INVALID;
    }
    return RSX_OK;
}

void *ws_take(rsx_engine *e, size_t bytes)
{
    size_t at = align_up(e->ws_used, 256);
    if (!e->arena || at + bytes > e->arena_size) return nullptr;
    e->ws_used = at + bytes;
    return e->arena + at;
}

namespace {

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// argument errors are reported before the engine's lock is taken: take it for the message
int bad(rsx_engine *e, const char *msg)
{
    if (e) {
        std::lock_guard<std::mutex> lk(e->mu);
        e->err = msg;
    }
    return RSX_E_INVALID;
}

bool format_ok(int f) { return f >= RSX_FORMAT_DEFLATE && f <= RSX_FORMAT_GZIP; }

void time_begin(rsx_engine *e) { (void)hipEventRecord(e->ev_begin, e->stream); }
void time_end(rsx_engine *e)
{
    (void)hipEventRecord(e->ev_end, e->stream);
    e->timed = true;
}

// host-side view of a batch: validates offsets and derives the scalars the
// device entry points want
struct BatchShape {
    uint64_t total = 0, max_item = 0;
};

bool shape_of(const uint64_t *off, size_t n, BatchShape *s)
{
    s->total = n ? off[n] - off[0] : 0;
    s->max_item = 0;
    for (size_t i = 0; i < n; i++) {
        if (off[i + 1] < off[i]) return false;
        uint64_t l = off[i + 1] - off[i];
        if (l > s->max_item) s->max_item = l;
    }
    return true;
}

}  // namespace

extern "C" {

int rsx_engine_create(int device, void *stream, rsx_engine **out)
{
    if (!out) return RSX_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return RSX_E_DEVICE;
    rsx_engine *e = new (std::nothrow) rsx_engine();
    if (!e) return RSX_E_NOMEM;
    e->device = device;
    DeviceGuard g(device);
    if (stream) {
        e->stream = static_cast<hipStream_t>(stream);
    } else {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
            delete e;
            return RSX_E_DEVICE;
        }
        e->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipEventCreate(&e->ev_begin) != hipSuccess || hipEventCreate(&e->ev_end) != hipSuccess ||
        checksum_tables_create(e) != RSX_OK || inflate_static_scratch(e) != RSX_OK ||
        hipGetDeviceProperties(&prop, device) != hipSuccess) {
        rsx_engine_destroy(e);
        return RSX_E_DEVICE;
    }
    e->num_cus = prop.multiProcessorCount;
    e->inflate_stats = getenv("RSX_INFLATE_STATS") != nullptr;
    e->inflate_split = getenv("RSX_INFLATE_FUSED") == nullptr && !e->inflate_stats;
    e->inflate_parts = getenv("RSX_INFLATE_WHOLE") == nullptr;
    *out = e;
    return RSX_OK;
}

void rsx_engine_destroy(rsx_engine *e)
{
    if (!e) return;
    DeviceGuard g(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->ws) (void)hipFree(e->ws);
    if (e->d_crc_tables) (void)hipFree(e->d_crc_tables);
    if (e->ev_begin) (void)hipEventDestroy(e->ev_begin);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

const char *rsx_engine_last_error(const rsx_engine *e) { return e ? e->err.c_str() : "null engine"; }

void *rsx_engine_stream(const rsx_engine *e) { return e ? e->stream : nullptr; }

int rsx_engine_synchronize(rsx_engine *e)
{
    if (!e) return RSX_E_INVALID;
    DeviceGuard g(e->device);
    RSX_HIP(e, hipStreamSynchronize(e->stream));
    return RSX_OK;
}

float rsx_engine_last_kernel_ms(rsx_engine *e)
{
    if (!e || !e->timed) return -1.0f;
    DeviceGuard g(e->device);
    float ms = -1.0f;
    if (hipEventSynchronize(e->ev_end) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, e->ev_begin, e->ev_end) != hipSuccess) return -1.0f;
    return ms;
}

static bool last_part_words(rsx_engine *e, uint32_t v[3])
{
    v[0] = v[1] = v[2] = 0;
    if (!e) return false;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->d_last_parts) return false;
    DeviceGuard g(e->device);
    if (hipStreamSynchronize(e->stream) != hipSuccess) return false;
    return hipMemcpy(v, e->d_last_parts, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess;
}

uint32_t rsx_engine_last_parts(rsx_engine *e)
{
    uint32_t v[3];
    return last_part_words(e, v) ? v[0] : 0;
}

uint32_t rsx_engine_last_whole(rsx_engine *e)
{
    uint32_t v[3];
    return last_part_words(e, v) ? v[1] + v[2] : 0;
}

uint32_t rsx_engine_last_fallbacks(rsx_engine *e)
{
    if (!e) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->d_last_fallbacks) return 0;
    DeviceGuard g(e->device);
    uint32_t v = 0;
    if (hipStreamSynchronize(e->stream) != hipSuccess) return 0;
    return hipMemcpy(&v, e->d_last_fallbacks, sizeof v, hipMemcpyDeviceToHost) == hipSuccess ? v : 0;
}

uint64_t rsx_compress_bound(int format, uint64_t len)
{
    uint64_t b = len + (len / 65535 + 1) * 5 + 10;
    if (format == RSX_FORMAT_ZLIB) b += 6;
    if (format == RSX_FORMAT_GZIP) b += 18;
    return b;
}

// ---- device-resident entry points -------------------------------------------

int rsx_batch_checksum_device(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                              const uint64_t *in_off, size_t n, uint64_t total_in,
                              uint64_t max_item, uint32_t *values)
{
    if (!e) return RSX_E_INVALID;
    if (kind != RSX_CHECKSUM_CRC32 && kind != RSX_CHECKSUM_ADLER32) return bad(e, "bad checksum kind");
    if (n == 0) return RSX_OK;
    if (!in_off || !values || (!in && total_in)) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    int r = ws_reserve(e, checksum_ws_bytes(n, total_in));
    if (r != RSX_OK) return r;
    time_begin(e);
    r = checksum_launch(e, kind, init, in, in_off, nullptr, n, total_in, max_item, values, false);
    time_end(e);
    return r;
}

int rsx_batch_decompress_device(rsx_engine *e, int format, const uint8_t *in,
                                const uint64_t *in_off, size_t n, uint64_t total_in,
                                uint64_t total_out, uint64_t max_out, uint8_t *out,
                                const uint64_t *out_off, uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!format_ok(format)) return bad(e, "bad format");
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    int r = ws_reserve(e, inflate_ws_bytes(n, total_in, total_out, e->inflate_split));
    if (r != RSX_OK) return r;
    return inflate_launch(e, format, in, in_off, n, total_in, total_out, max_out, out, out_off,
                          out_sizes, status, e->inflate_split);
}

int rsx_batch_decompress_device_noalloc(rsx_engine *e, int format, const uint8_t *in,
                                        const uint64_t *in_off, size_t n, uint64_t total_in,
                                        uint64_t total_out, uint64_t max_out, uint8_t *out,
                                        const uint64_t *out_off, uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!format_ok(format)) return bad(e, "bad format");
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    if (n > RSX_NOALLOC_MAX_ITEMS || total_out > RSX_NOALLOC_MAX_TOTAL_OUT)
        return bad(e, "batch too large for the static scratch (n <= 8192 items, total_out <= 1 GiB)");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    // the one-kernel decoder: its scratch (16 bytes per item and the checksum partials) fits the
    // static array; the token arrays of the two-kernel decoder do not
    if (ws_use_static(e, inflate_ws_bytes(n, total_in, total_out, false)) != RSX_OK) {
        e->err = "static scratch too small for this batch";
        return RSX_E_INVALID;
    }
    return inflate_launch(e, format, in, in_off, n, total_in, total_out, max_out, out, out_off,
                          out_sizes, status, false);
}

int rsx_batch_compress_device(rsx_engine *e, int format, int level, const uint8_t *in,
                              const uint64_t *in_off, size_t n, uint64_t total_in,
                              uint64_t max_item, uint8_t *out, const uint64_t *out_off,
                              uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!format_ok(format)) return bad(e, "bad format");
    if (level < 0 || level > 12) return bad(e, "Compression level must be between 0 and 12");
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    int r = ws_reserve(e, deflate_ws_bytes(n, total_in, max_item, level));
    if (r != RSX_OK) return r;
    return deflate_launch(e, format, level, in, in_off, n, total_in, max_item, out, out_off,
                          out_sizes, status, RSX_FLUSH_FINISH, false);
}

// ---- host-pointer entry points ------------------------------------------------

namespace {

// Device staging for one host call: flat input, both offset arrays (rebased so
// that item 0 starts at 0), flat output, sizes, status.
struct Staged {
    uint8_t *d_in = nullptr, *d_out = nullptr;
    uint64_t *d_in_off = nullptr, *d_out_off = nullptr, *d_sizes = nullptr;
    int32_t *d_status = nullptr;
    uint32_t *d_values = nullptr;
};

}  // namespace

static int stage_and_run(rsx_engine *e, int op, int format, int level, int kind, uint32_t init,
                         const uint8_t *in, const uint64_t *in_off, size_t n, uint8_t *out,
                         const uint64_t *out_off, uint64_t *out_sizes, int32_t *status,
                         uint32_t *values)
{
    BatchShape si, so;
    if (!shape_of(in_off, n, &si)) return bad(e, "in_off not monotonic");
    if (out_off && !shape_of(out_off, n, &so)) return bad(e, "out_off not monotonic");
    if ((!in && si.total) || (out_off && !out && so.total)) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);

    size_t kernel_ws = op == 0   ? deflate_ws_bytes(n, si.total, si.max_item, level)
                       : op == 1 ? inflate_ws_bytes(n, si.total, so.total, e->inflate_split)
                                 : checksum_ws_bytes(n, si.total);
    size_t need = kernel_ws + align_up(si.total + 64, 256) + align_up(so.total + 64, 256) +
                  4 * align_up((n + 1) * sizeof(uint64_t), 256) + 2 * align_up(n * 4, 256) + 4096;
    int r = ws_reserve(e, need);
    if (r != RSX_OK) return r;

    Staged s;
    s.d_in = static_cast<uint8_t *>(ws_take(e, si.total + 64));
    s.d_in_off = static_cast<uint64_t *>(ws_take(e, (n + 1) * sizeof(uint64_t)));
    if (out_off) {
        s.d_out = static_cast<uint8_t *>(ws_take(e, so.total + 64));
        s.d_out_off = static_cast<uint64_t *>(ws_take(e, (n + 1) * sizeof(uint64_t)));
        s.d_sizes = static_cast<uint64_t *>(ws_take(e, n * sizeof(uint64_t)));
        s.d_status = static_cast<int32_t *>(ws_take(e, n * sizeof(int32_t)));
    } else {
        s.d_values = static_cast<uint32_t *>(ws_take(e, n * sizeof(uint32_t)));
    }

    if (!s.d_in || !s.d_in_off || (out_off && (!s.d_out || !s.d_out_off || !s.d_sizes || !s.d_status)) ||
        (!out_off && !s.d_values)) {
        e->err = "workspace carve failed";
        return RSX_E_NOMEM;
    }

    // rebase offsets to the staged copies
    uint64_t *h_off = new (std::nothrow) uint64_t[2 * (n + 1)];
    if (!h_off) {
        e->err = "host allocation of the offset arrays failed";
        return RSX_E_NOMEM;
    }
    for (size_t i = 0; i <= n; i++) h_off[i] = in_off[i] - in_off[0];
    if (out_off)
        for (size_t i = 0; i <= n; i++) h_off[n + 1 + i] = out_off[i] - out_off[0];
    hipError_t st = hipSuccess;
    if (si.total)
        st = hipMemcpyAsync(s.d_in, in + in_off[0], si.total, hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess)
        st = hipMemcpyAsync(s.d_in_off, h_off, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice,
                            e->stream);
    if (st == hipSuccess && out_off)
        st = hipMemcpyAsync(s.d_out_off, h_off + n + 1, (n + 1) * sizeof(uint64_t),
                            hipMemcpyHostToDevice, e->stream);
    if (st == hipSuccess) st = hipStreamSynchronize(e->stream);
    delete[] h_off;
    if (st != hipSuccess) {
        e->err = std::string("staging copy: ") + hipGetErrorString(st);
        return RSX_E_DEVICE;
    }

    if (op == 0)
        r = deflate_launch(e, format, level, s.d_in, s.d_in_off, n, si.total, si.max_item, s.d_out,
                           s.d_out_off, s.d_sizes, s.d_status, kind /* compress: the flush mode */, true);
    else if (op == 1)
        r = inflate_launch(e, format, s.d_in, s.d_in_off, n, si.total, so.total, so.max_item,
                           s.d_out, s.d_out_off, s.d_sizes, s.d_status, e->inflate_split);
    else
        r = checksum_launch(e, kind, init, s.d_in, s.d_in_off, nullptr, n, si.total, si.max_item,
                            s.d_values, true);
    if (r != RSX_OK) return r;

    if (out_off) {
        RSX_HIP(e, hipMemcpyAsync(out_sizes, s.d_sizes, n * sizeof(uint64_t), hipMemcpyDeviceToHost,
                                  e->stream));
        RSX_HIP(e, hipMemcpyAsync(status, s.d_status, n * sizeof(int32_t), hipMemcpyDeviceToHost,
                                  e->stream));
        RSX_HIP(e, hipStreamSynchronize(e->stream));
        // copy back only what each item produced; sizes are validated against the
        // slot like src/batch_cuda.rs:123-137 does before touching the payload
        for (size_t i = 0; i < n; i++) {
            if (out_sizes[i] > out_off[i + 1] - out_off[i]) {
                e->err = "device returned a size larger than its slot";
                return RSX_E_DEVICE;
            }
        }
        // items that fill their slots to the last byte (decompression into exact-size slots)
        // are adjacent in both buffers: one copy per run of them, not one per item
        uint64_t run_at = 0, run_len = 0;   // pending copy, relative to out_off[0]
        for (size_t i = 0; i <= n; i++) {
            const bool have = i < n && status[i] == 0 && out_sizes[i];
            const uint64_t at = i < n ? out_off[i] - out_off[0] : 0;
            if (run_len && (!have || at != run_at + run_len)) {
                RSX_HIP(e, hipMemcpyAsync(out + out_off[0] + run_at, s.d_out + run_at, run_len,
                                          hipMemcpyDeviceToHost, e->stream));
                run_len = 0;
            }
            if (have) {
                if (!run_len) run_at = at;
                run_len += out_sizes[i];
            }
        }
        RSX_HIP(e, hipStreamSynchronize(e->stream));
    } else {
        RSX_HIP(e, hipMemcpyAsync(values, s.d_values, n * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                  e->stream));
        RSX_HIP(e, hipStreamSynchronize(e->stream));
    }
    return RSX_OK;
}

int rsx_batch_compress(rsx_engine *e, int format, int level, const uint8_t *in,
                       const uint64_t *in_off, size_t n, uint8_t *out, const uint64_t *out_off,
                       uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!format_ok(format)) return bad(e, "bad format");
    if (level < 0 || level > 12) return bad(e, "Compression level must be between 0 and 12");
    if (n == 0) return RSX_OK;
    if (!in_off || !out || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    return stage_and_run(e, 0, format, level, RSX_FLUSH_FINISH, 0, in, in_off, n, out, out_off, out_sizes,
                         status, nullptr);
}

int rsx_batch_compress_flush(rsx_engine *e, int level, int flush, const uint8_t *in,
                             const uint64_t *in_off, size_t n, uint8_t *out, const uint64_t *out_off,
                             uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (flush != RSX_FLUSH_SYNC && flush != RSX_FLUSH_FINISH) return bad(e, "bad flush mode");
    if (level < 0 || level > 12) return bad(e, "Compression level must be between 0 and 12");
    if (n == 0) return RSX_OK;
    if (!in_off || !out || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    return stage_and_run(e, 0, RSX_FORMAT_DEFLATE, level, flush, 0, in, in_off, n, out, out_off, out_sizes,
                         status, nullptr);
}

int rsx_batch_decompress(rsx_engine *e, int format, const uint8_t *in, const uint64_t *in_off,
                         size_t n, uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                         int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!format_ok(format)) return bad(e, "bad format");
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    return stage_and_run(e, 1, format, 0, 0, 0, in, in_off, n, out, out_off, out_sizes, status,
                         nullptr);
}

int rsx_batch_decompress_members(rsx_engine *e, const uint8_t *in, const uint64_t *in_off, size_t n,
                                 uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                                 int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    BatchShape si, so;
    if (!shape_of(in_off, n, &si)) return bad(e, "in_off not monotonic");
    if (!shape_of(out_off, n, &so)) return bad(e, "out_off not monotonic");
    if ((!in && si.total) || (!out && so.total)) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    const size_t arr = align_up((n + 1) * sizeof(uint64_t), 256);
    const size_t need = inflate_ws_bytes(n, si.total, so.total, e->inflate_split) + align_up(si.total + 64, 256) +
                        align_up(so.total + 64, 256) + 8 * arr + 4096;
    int r = ws_reserve(e, need);
    if (r != RSX_OK) return r;
    uint8_t *d_in = static_cast<uint8_t *>(ws_take(e, si.total + 64));
    uint8_t *d_out = static_cast<uint8_t *>(ws_take(e, so.total + 64));
    uint64_t *d_lo = static_cast<uint64_t *>(ws_take(e, arr)), *d_hi = static_cast<uint64_t *>(ws_take(e, arr));
    uint64_t *d_olo = static_cast<uint64_t *>(ws_take(e, arr)), *d_ohi = static_cast<uint64_t *>(ws_take(e, arr));
    uint64_t *d_used = static_cast<uint64_t *>(ws_take(e, arr)), *d_sizes = static_cast<uint64_t *>(ws_take(e, arr));
    uint64_t *d_base = static_cast<uint64_t *>(ws_take(e, 256));
    int32_t *d_status = static_cast<int32_t *>(ws_take(e, arr));
    if (!d_in || !d_out || !d_lo || !d_hi || !d_olo || !d_ohi || !d_used || !d_sizes || !d_base || !d_status) {
        e->err = "workspace carve failed";
        return RSX_E_NOMEM;
    }
    const size_t mark = e->ws_used;   // what the rounds carve lies behind this
    const uint8_t *src = in + in_off[0];
    if (si.total) RSX_HIP(e, hipMemcpyAsync(d_in, src, si.total, hipMemcpyHostToDevice, e->stream));
    RSX_HIP(e, hipMemsetAsync(d_base, 0, sizeof(uint64_t), e->stream));

    // per item: where its next member starts and where its next output byte goes (relative to the staged copies)
    std::vector<uint64_t> cur_in(n), cur_out(n), h(4 * n), h_used(n), h_sizes(n);
    std::vector<int32_t> h_status(n);
    std::vector<size_t> active(n), next;
    for (size_t i = 0; i < n; i++) {
        cur_in[i] = in_off[i] - in_off[0];
        cur_out[i] = out_off[i] - out_off[0];
        out_sizes[i] = 0;
        status[i] = RSX_SUCCESS;
        active[i] = i;
    }
    // One launch chain per member depth.  An item of thousands of tiny members would hold the engine for as many
    // serial launches while every other item waits: items still unfinished after kMaxMemberRounds members are
    // reported RSX_BAD_DATA (gzip sets no limit; pigz and concatenated archives stay far below this one).
    constexpr size_t kMaxMemberRounds = 4096;
    size_t rounds_done = 0;
    auto rounds = [&]() -> int {
    while (!active.empty()) {
        if (rounds_done++ == kMaxMemberRounds) {
            for (size_t i : active) {
                status[i] = RSX_BAD_DATA;
                out_sizes[i] = 0;
            }
            active.clear();
            break;
        }
        const size_t m = active.size();
        uint64_t max_out = 0;
        for (size_t k = 0; k < m; k++) {
            const size_t i = active[k];
            h[k] = cur_in[i];
            h[m + k] = in_off[i + 1] - in_off[0];
            h[2 * m + k] = cur_out[i];
            h[3 * m + k] = out_off[i + 1] - out_off[0];
            if (h[3 * m + k] - h[2 * m + k] > max_out) max_out = h[3 * m + k] - h[2 * m + k];
        }
        RSX_HIP(e, hipMemcpyAsync(d_lo, h.data(), m * 8, hipMemcpyHostToDevice, e->stream));
        RSX_HIP(e, hipMemcpyAsync(d_hi, h.data() + m, m * 8, hipMemcpyHostToDevice, e->stream));
        RSX_HIP(e, hipMemcpyAsync(d_olo, h.data() + 2 * m, m * 8, hipMemcpyHostToDevice, e->stream));
        RSX_HIP(e, hipMemcpyAsync(d_ohi, h.data() + 3 * m, m * 8, hipMemcpyHostToDevice, e->stream));
        e->ws_used = mark;
        const InflateSlices sl{d_hi, d_ohi, d_base, d_used};
        r = inflate_launch(e, RSX_FORMAT_GZIP, d_in, d_lo, m, si.total, so.total, max_out, d_out, d_olo, d_sizes,
                           d_status, e->inflate_split, &sl);
        if (r != RSX_OK) return r;
        RSX_HIP(e, hipMemcpyAsync(h_sizes.data(), d_sizes, m * 8, hipMemcpyDeviceToHost, e->stream));
        RSX_HIP(e, hipMemcpyAsync(h_used.data(), d_used, m * 8, hipMemcpyDeviceToHost, e->stream));
        RSX_HIP(e, hipMemcpyAsync(h_status.data(), d_status, m * 4, hipMemcpyDeviceToHost, e->stream));
        RSX_HIP(e, hipStreamSynchronize(e->stream));
        next.clear();
        for (size_t k = 0; k < m; k++) {
            const size_t i = active[k];
            const uint64_t end = in_off[i + 1] - in_off[0];
            if (h_status[k] != RSX_SUCCESS || h_used[k] == 0 || h_used[k] > end - cur_in[i] ||
                h_sizes[k] > (out_off[i + 1] - out_off[0]) - cur_out[i]) {
                status[i] = h_status[k] != RSX_SUCCESS ? h_status[k] : RSX_BAD_DATA;
                out_sizes[i] = 0;
                continue;
            }
            out_sizes[i] += h_sizes[k];
            cur_out[i] += h_sizes[k];
            cur_in[i] += h_used[k];
            // what follows the member: nothing, zero padding, or another member
            bool zeros = true;
            for (uint64_t q = cur_in[i]; q < end && zeros; q++) zeros = src[q] == 0;
            if (zeros) continue;
            if (end - cur_in[i] >= 18 && src[cur_in[i]] == 0x1F && src[cur_in[i] + 1] == 0x8B) {
                next.push_back(i);
            } else {
                status[i] = RSX_BAD_DATA;
                out_sizes[i] = 0;
            }
        }
        active.swap(next);
    }
    return RSX_OK;
    };
    r = rounds();
    if (r != RSX_OK) {
        // an engine error in the middle: nothing of the items still in flight is usable
        for (size_t i : active) {
            status[i] = RSX_INTERNAL_ERROR;
            out_sizes[i] = 0;
        }
        return r;
    }
    for (size_t i = 0; i < n; i++)
        if (status[i] == RSX_SUCCESS && out_sizes[i])
            RSX_HIP(e, hipMemcpyAsync(out + out_off[i], d_out + (out_off[i] - out_off[0]), out_sizes[i],
                                      hipMemcpyDeviceToHost, e->stream));
    RSX_HIP(e, hipStreamSynchronize(e->stream));
    return RSX_OK;
}

int rsx_batch_checksum(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                       const uint64_t *in_off, size_t n, uint32_t *values)
{
    if (!e) return RSX_E_INVALID;
    if (kind != RSX_CHECKSUM_CRC32 && kind != RSX_CHECKSUM_ADLER32) return bad(e, "bad checksum kind");
    if (n == 0) return RSX_OK;
    if (!in_off || !values) return bad(e, "null pointer");
    return stage_and_run(e, 2, 0, 0, kind, init, in, in_off, n, nullptr, nullptr, nullptr, nullptr,
                         values);
}

int rsx_crc32(rsx_engine *e, uint32_t crc, const uint8_t *data, size_t len, uint32_t *value)
{
    uint64_t off[2] = {0, len};
    return rsx_batch_checksum(e, RSX_CHECKSUM_CRC32, crc, data, off, 1, value);
}

int rsx_adler32(rsx_engine *e, uint32_t adler, const uint8_t *data, size_t len, uint32_t *value)
{
    uint64_t off[2] = {0, len};
    return rsx_batch_checksum(e, RSX_CHECKSUM_ADLER32, adler, data, off, 1, value);
}

}  // extern "C"
