// api.cpp -- the extern "C" surface declared in include/rsx_deflate.h: engine
// lifetime, workspace management, and the host-pointer wrappers that stage
// through device memory the way src/batch_cuda.rs:81-141 does (H2D, launch,
// D2H) around the *_device entry points -- slice by slice, so that the copy
// in of one slice, the kernels of the one before and the copy out of the one
// before that run side by side -- and, for engines bound to several GPUs, one
// contiguous range of the batch per GPU.
#include "engine.h"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <thread>
#include <vector>

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

// Threads that copy between pinned bounce buffers and the caller's memory.  run() hands out the
// indices [0, count) to the pool's threads and to the caller and returns when all are done.
class WorkerPool {
public:
    explicit WorkerPool(unsigned threads)
    {
        for (unsigned i = 0; i < threads; i++) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(size_t count, const std::function<void(size_t)> &fn)
    {
        if (count == 0) return;
        if (th_.empty() || count == 1) {
            for (size_t i = 0; i < count; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            count_ = count;
            next_.store(0);
            active_ = th_.size();
            gen_++;
        }
        cv_.notify_all();
        for (size_t i; (i = next_.fetch_add(1)) < count;) fn(i);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return active_ == 0; });
        fn_ = nullptr;
    }

private:
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)> *fn;
            size_t count;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                fn = fn_;
                count = count_;
            }
            for (size_t i; (i = next_.fetch_add(1)) < count;) (*fn)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--active_ == 0) done_cv_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t count_ = 0, active_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool stop_ = false;
};

// What the host entry points keep between calls: a stream per copy direction (the engine's own stream
// runs the kernels), two pinned buffers the packed results of a slice land in, two pinned blocks for the
// slices' small arrays, and the copy threads.
struct HostPipe {
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    uint8_t *pin_out[2] = {nullptr, nullptr};
    size_t pin_out_size = 0;
    uint8_t *pin_meta[2] = {nullptr, nullptr};
    size_t pin_meta_size = 0;
    WorkerPool *pool = nullptr;
};

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

int single_only(rsx_engine *e)
{
    return bad(e, "device pointers belong to one GPU: the *_device entry points need an engine bound to a single device");
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

void pipe_destroy(rsx_engine *e)
{
    HostPipe *p = e->pipe;
    if (!p) return;
    delete p->pool;
    for (int k = 0; k < 2; k++) {
        if (p->pin_out[k]) (void)hipHostFree(p->pin_out[k]);
        if (p->pin_meta[k]) (void)hipHostFree(p->pin_meta[k]);
        if (p->ev_in[k]) (void)hipEventDestroy(p->ev_in[k]);
        if (p->ev_k[k]) (void)hipEventDestroy(p->ev_k[k]);
        if (p->ev_out[k]) (void)hipEventDestroy(p->ev_out[k]);
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
    e->pipe = nullptr;
}

unsigned env_uint(const char *name, unsigned dflt, unsigned lo, unsigned hi)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const long x = strtol(v, nullptr, 10);
    return x < (long)lo ? lo : x > (long)hi ? hi : (unsigned)x;
}

int pipe_create(rsx_engine *e)
{
    if (e->pipe) return RSX_OK;
    HostPipe *p = new (std::nothrow) HostPipe();
    if (!p) return RSX_E_NOMEM;
    e->pipe = p;
    RSX_HIP(e, hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    RSX_HIP(e, hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    for (int k = 0; k < 2; k++) {
        RSX_HIP(e, hipEventCreateWithFlags(&p->ev_in[k], hipEventDisableTiming));
        RSX_HIP(e, hipEventCreateWithFlags(&p->ev_k[k], hipEventDisableTiming));
        RSX_HIP(e, hipEventCreateWithFlags(&p->ev_out[k], hipEventDisableTiming));
    }
    // RSX_HOST_THREADS: copy threads per engine beside the calling one (0: the caller copies alone)
    p->pool = new (std::nothrow) WorkerPool(env_uint("RSX_HOST_THREADS", 7, 0, 63));
    return p->pool ? RSX_OK : RSX_E_NOMEM;
}

int pipe_pinned(rsx_engine *e, size_t out_bytes, size_t meta_bytes)
{
    HostPipe *p = e->pipe;
    if (out_bytes > p->pin_out_size) {
        for (int k = 0; k < 2; k++) {
            if (p->pin_out[k]) (void)hipHostFree(p->pin_out[k]);
            p->pin_out[k] = nullptr;
        }
        p->pin_out_size = 0;
        const size_t want = align_up(out_bytes, 1 << 20);
        for (int k = 0; k < 2; k++) RSX_HIP(e, hipHostMalloc(reinterpret_cast<void **>(&p->pin_out[k]), want, hipHostMallocDefault));
        p->pin_out_size = want;
    }
    if (meta_bytes > p->pin_meta_size) {
        for (int k = 0; k < 2; k++) {
            if (p->pin_meta[k]) (void)hipHostFree(p->pin_meta[k]);
            p->pin_meta[k] = nullptr;
        }
        p->pin_meta_size = 0;
        const size_t want = align_up(meta_bytes, 1 << 16);
        for (int k = 0; k < 2; k++) RSX_HIP(e, hipHostMalloc(reinterpret_cast<void **>(&p->pin_meta[k]), want, hipHostMallocDefault));
        p->pin_meta_size = want;
    }
    return RSX_OK;
}

// ---- packing a slice's results on the device ----------------------------------------------
// The items' slots are as large as they may get; what the items produced is usually less (compressed
// sizes against the bound, failed items).  Before a slice goes back over PCIe its results are packed
// back to back, in item order, so that only produced bytes travel.

constexpr uint32_t kPackTile = 1u << 16;   // packed bytes per workgroup

// exclusive prefix sums of the produced sizes (one workgroup); pack_off[m] = the slice's packed bytes
__global__ __launch_bounds__(1024) void pack_scan_kernel(const uint64_t *__restrict__ sizes, const int32_t *__restrict__ status,
                                                         uint32_t m, uint64_t *__restrict__ pack_off)
{
    __shared__ uint64_t wsum[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    WG_SYNC();
    for (uint32_t base = 0; base < m; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t v = i < m && status[i] == 0 ? sizes[i] : 0;
        uint64_t x = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = __shfl_up(x, d);
            if ((int)(threadIdx.x & 63) >= d) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        WG_SYNC();
        uint64_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < m) pack_off[i] = before + x - v;
        WG_SYNC();
        if (threadIdx.x == 1023) carry = before + x;
        WG_SYNC();
    }
    if (threadIdx.x == 0) pack_off[m] = carry;
}

// All lanes of a workgroup: count bytes from src to dst, any alignment on both sides: the destination
// is written as aligned dwords made of two aligned source dwords.
__device__ inline void wg_copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t count)
{
    const uint32_t tid = threadIdx.x, nt = blockDim.x;
    uint64_t head = (4 - ((uintptr_t)dst & 3)) & 3;
    if (head > count) head = count;
    if (tid < head) dst[tid] = src[tid];
    const uint64_t ndw = (count - head) >> 2;
    const uint8_t *s = src + head;
    const uint32_t sh = (uint32_t)((uintptr_t)s & 3);
    const uint32_t *sa = reinterpret_cast<const uint32_t *>(s - sh);
    uint32_t *da = reinterpret_cast<uint32_t *>(dst + head);
    if (sh == 0) {
        for (uint64_t j = tid; j < ndw; j += nt) da[j] = sa[j];
    } else {
        // (the dword behind the last one needed holds at least one byte of the range: no read past it)
        for (uint64_t j = tid; j < ndw; j += nt) da[j] = __builtin_amdgcn_alignbyte(sa[j + 1], sa[j], sh);
    }
    const uint64_t done = head + 4 * ndw;
    if (tid < count - done) dst[done + tid] = src[done + tid];
}

// one workgroup per kPackTile bytes of the packed buffer: the pieces of the items that fall into its tile
__global__ __launch_bounds__(256) void pack_copy_kernel(const uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
                                                        const uint64_t *__restrict__ pack_off, uint32_t m,
                                                        uint8_t *__restrict__ pack)
{
    const uint64_t total = pack_off[m];
    const uint64_t lo = (uint64_t)blockIdx.x * kPackTile;
    if (lo >= total) return;
    const uint64_t hi = lo + kPackTile < total ? lo + kPackTile : total;
    uint32_t a = 0, b = m;   // the last item with pack_off[item] <= lo
    while (b - a > 1) {
        const uint32_t mid = a + (b - a) / 2;
        if (pack_off[mid] <= lo) a = mid; else b = mid;
    }
    for (uint32_t i = a; i < m && pack_off[i] < hi; i++) {
        const uint64_t p0 = pack_off[i], p1 = pack_off[i + 1];
        const uint64_t from = p0 > lo ? p0 : lo, to = p1 < hi ? p1 : hi;
        if (to > from) wg_copy_bytes(pack + from, out + (out_off[i] - out_off[0]) + (from - p0), to - from);
    }
}

// a slice's diagnostic words (any may be null) added to the call's
__global__ void diag_add_kernel(const uint32_t *parts, const uint32_t *fallbacks, uint32_t *acc)
{
    if (parts) {
        acc[0] += parts[0];
        acc[1] += parts[1];
        acc[2] += parts[2];
    }
    if (fallbacks) acc[3] += fallbacks[0];
}

// A slice: a run of consecutive items that is staged, processed and copied back as one piece.
struct Slice {
    size_t i0 = 0, i1 = 0;
    uint64_t in_bytes = 0, out_bytes = 0, max_in = 0, max_out = 0;
};

constexpr size_t kMaxSliceItems = 1u << 16;

std::vector<Slice> plan_slices(const uint64_t *in_off, const uint64_t *out_off, size_t n, uint64_t slice_bytes)
{
    std::vector<Slice> v;
    Slice s;
    for (size_t i = 0; i < n; i++) {
        const uint64_t il = in_off[i + 1] - in_off[i], ol = out_off ? out_off[i + 1] - out_off[i] : 0;
        if (s.i1 > s.i0 && (s.in_bytes + il > slice_bytes || s.out_bytes + ol > slice_bytes || s.i1 - s.i0 >= kMaxSliceItems)) {
            v.push_back(s);
            s = Slice();
            s.i0 = s.i1 = i;
        }
        s.i1 = i + 1;
        s.in_bytes += il;
        s.out_bytes += ol;
        if (il > s.max_in) s.max_in = il;
        if (ol > s.max_out) s.max_out = ol;
    }
    if (s.i1 > s.i0) v.push_back(s);
    return v;
}

// what one parity of the pinned meta block holds for a slice of m items
struct MetaLayout {
    size_t in_off, out_off, sizes, status, values, total;
    explicit MetaLayout(size_t m)
    {
        in_off = 0;
        out_off = in_off + align_up((m + 1) * 8, 64);
        sizes = out_off + align_up((m + 1) * 8, 64);
        status = sizes + align_up(m * 8, 64);
        values = status + align_up(m * 4, 64);
        total = values + align_up(m * 4, 64);
    }
};

}  // namespace

// One host call on one GPU.  op: 0 compress (kind = the flush mode), 1 decompress, 2 checksum.
//
// The batch is cut into slices of about RSX_HOST_SLICE_MB (128) MiB of input or of output slots.  Slice k is
// copied to the device by a helper thread (straight from the caller's memory: that copy is the driver's), its
// kernels run on the engine's stream, its results are packed back to back on the device and copied into a
// pinned buffer on a third stream, and the calling thread and the pool's copy threads move them from there
// into the caller's slots -- while slice k + 1 is on the device and slice k + 2 on its way in.  (One
// synchronous copy in, launch, copy out through pageable memory, as src/batch_cuda.rs:81-141 does it,
// is bound by the copy out: 17 GB/s on the test machine.)
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
    int r = pipe_create(e);
    if (r != RSX_OK) return r;
    HostPipe *const P = e->pipe;

    const uint64_t slice_bytes = (uint64_t)env_uint("RSX_HOST_SLICE_MB", 128, 1, 4096) << 20;
    const std::vector<Slice> slices = plan_slices(in_off, out_off, n, slice_bytes);
    const size_t ns = slices.size();
    uint64_t max_in = 0, max_out = 0;
    size_t max_m = 0, kernel_ws = 0;
    for (const Slice &s : slices) {
        const size_t m = s.i1 - s.i0;
        if (s.in_bytes > max_in) max_in = s.in_bytes;
        if (s.out_bytes > max_out) max_out = s.out_bytes;
        if (m > max_m) max_m = m;
        const size_t w = op == 0   ? deflate_ws_bytes(m, s.in_bytes, s.max_in, level)
                         : op == 1 ? inflate_ws_bytes(m, s.in_bytes, s.out_bytes, e->inflate_split)
                                   : checksum_ws_bytes(m, s.in_bytes);
        if (w > kernel_ws) kernel_ws = w;
    }
    const size_t arr8 = align_up((max_m + 1) * 8, 256), arr4 = align_up(max_m * 4 + 4, 256);
    const size_t need = kernel_ws + 2 * align_up(max_in + 64, 256) + 3 * align_up(max_out + 64, 256) + 4 * arr8 + 2 * arr4 + 8192;
    r = ws_reserve(e, need);
    if (r != RSX_OK) return r;

    uint8_t *d_in[2], *d_pack[2] = {nullptr, nullptr};
    d_in[0] = static_cast<uint8_t *>(ws_take(e, max_in + 64));
    d_in[1] = static_cast<uint8_t *>(ws_take(e, max_in + 64));
    uint64_t *d_in_off = static_cast<uint64_t *>(ws_take(e, (max_m + 1) * 8));
    uint8_t *d_out = nullptr;
    uint64_t *d_out_off = nullptr, *d_sizes = nullptr, *d_pack_off = nullptr;
    int32_t *d_status = nullptr;
    uint32_t *d_values = nullptr;
    if (out_off) {
        d_out = static_cast<uint8_t *>(ws_take(e, max_out + 64));
        d_pack[0] = static_cast<uint8_t *>(ws_take(e, max_out + 64));
        d_pack[1] = static_cast<uint8_t *>(ws_take(e, max_out + 64));
        d_out_off = static_cast<uint64_t *>(ws_take(e, (max_m + 1) * 8));
        d_sizes = static_cast<uint64_t *>(ws_take(e, max_m * 8 + 8));
        d_pack_off = static_cast<uint64_t *>(ws_take(e, (max_m + 1) * 8));
        d_status = static_cast<int32_t *>(ws_take(e, max_m * 4 + 4));
    } else {
        d_values = static_cast<uint32_t *>(ws_take(e, max_m * 4 + 4));
    }
    if (!d_in[0] || !d_in[1] || !d_in_off ||
        (out_off && (!d_out || !d_pack[0] || !d_pack[1] || !d_out_off || !d_sizes || !d_pack_off || !d_status)) ||
        (!out_off && !d_values)) {
        e->err = "workspace carve failed";
        return RSX_E_NOMEM;
    }
    const size_t mark = e->ws_used;   // what the slices' kernels carve lies behind this
    const MetaLayout ML(max_m);
    r = pipe_pinned(e, out_off ? (size_t)(max_out < slice_bytes ? max_out : slice_bytes) + 64 : 0, ML.total + 64);
    if (r != RSX_OK) return r;
    RSX_HIP(e, hipMemsetAsync(e->d_diag, 0, 4 * sizeof(uint32_t), e->stream));

    // ---- the copy in: a helper thread, one slice ahead of the kernels -------------------------
    std::mutex hm;
    std::condition_variable hcv;
    size_t in_issued = 0, launched = 0;   // slices whose ev_in / ev_k has been recorded (under hm)
    bool abort_all = false;
    hipError_t helper_err = hipSuccess;
    std::thread helper([&] {
        (void)hipSetDevice(e->device);
        for (size_t k = 0; k < ns; k++) {
            {
                // d_in[k & 1] is free once the kernels of slice k - 2 are done
                std::unique_lock<std::mutex> l(hm);
                hcv.wait(l, [&] { return abort_all || k < 2 || launched >= k - 1; });
                if (abort_all) return;
            }
            hipError_t st = hipSuccess;
            if (k >= 2) st = hipEventSynchronize(P->ev_k[k & 1]);
            const Slice &s = slices[k];
            if (st == hipSuccess && s.in_bytes)
                st = hipMemcpyAsync(d_in[k & 1], in + in_off[s.i0], s.in_bytes, hipMemcpyHostToDevice, P->s_in);
            if (st == hipSuccess) st = hipEventRecord(P->ev_in[k & 1], P->s_in);
            std::lock_guard<std::mutex> l(hm);
            if (st != hipSuccess) {
                helper_err = st;
                abort_all = true;
            }
            in_issued = k + 1;
            hcv.notify_all();
            if (st != hipSuccess) return;
        }
    });
    auto fail = [&](int code, const std::string &msg) {
        {
            std::lock_guard<std::mutex> l(hm);
            abort_all = true;
        }
        hcv.notify_all();
        helper.join();
        (void)hipStreamSynchronize(P->s_in);
        (void)hipStreamSynchronize(e->stream);
        (void)hipStreamSynchronize(P->s_out);
        if (!msg.empty()) e->err = msg;
        return code;
    };
#define PIPE_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t _st = (call);                                                         \
        if (_st != hipSuccess) return fail(RSX_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(_st)); \
    } while (0)

    std::vector<uint64_t> packed(ns, 0);        // packed bytes of every slice
    std::vector<uint8_t> direct(ns, 0);         // the slice goes back without the pinned buffer (too large for it)
    struct CopyJob {
        uint8_t *dst;
        const uint8_t *src;
        size_t len;
    };
    std::vector<CopyJob> jobs;

    for (size_t k = 0; k < ns + 2; k++) {
        // ---- launch slice k -----------------------------------------------------------------
        if (k < ns) {
            const Slice &s = slices[k];
            const size_t m = s.i1 - s.i0;
            uint8_t *meta = P->pin_meta[k & 1];
            {
                std::unique_lock<std::mutex> l(hm);
                hcv.wait(l, [&] { return abort_all || in_issued > k; });
                if (abort_all) {
                    l.unlock();
                    return fail(RSX_E_DEVICE, std::string("staging copy: ") + hipGetErrorString(helper_err));
                }
            }
            // the slice's offsets, rebased to its staged copies
            uint64_t *h_in_off = reinterpret_cast<uint64_t *>(meta + ML.in_off);
            uint64_t *h_out_off = reinterpret_cast<uint64_t *>(meta + ML.out_off);
            for (size_t i = 0; i <= m; i++) h_in_off[i] = in_off[s.i0 + i] - in_off[s.i0];
            PIPE_HIP(hipMemcpyAsync(d_in_off, h_in_off, (m + 1) * 8, hipMemcpyHostToDevice, e->stream));
            if (out_off) {
                for (size_t i = 0; i <= m; i++) h_out_off[i] = out_off[s.i0 + i] - out_off[s.i0];
                PIPE_HIP(hipMemcpyAsync(d_out_off, h_out_off, (m + 1) * 8, hipMemcpyHostToDevice, e->stream));
            }
            PIPE_HIP(hipStreamWaitEvent(e->stream, P->ev_in[k & 1], 0));
            if (k >= 2 && out_off) PIPE_HIP(hipStreamWaitEvent(e->stream, P->ev_out[k & 1], 0));   // d_pack[k & 1] is free
            e->ws_used = mark;
            if (op == 0)
                r = deflate_launch(e, format, level, d_in[k & 1], d_in_off, m, s.in_bytes, s.max_in, d_out, d_out_off,
                                   d_sizes, d_status, kind /* compress: the flush mode */, true);
            else if (op == 1)
                r = inflate_launch(e, format, d_in[k & 1], d_in_off, m, s.in_bytes, s.out_bytes, s.max_out, d_out,
                                   d_out_off, d_sizes, d_status, e->inflate_split);
            else
                r = checksum_launch(e, kind, init, d_in[k & 1], d_in_off, nullptr, m, s.in_bytes, s.max_in, d_values, true);
            if (r != RSX_OK) return fail(r, "");
            if (op == 1) diag_add_kernel<<<1, 1, 0, e->stream>>>(e->d_last_parts, e->d_last_fallbacks, e->d_diag);
            if (out_off) {
                pack_scan_kernel<<<1, 1024, 0, e->stream>>>(d_sizes, d_status, (uint32_t)m, d_pack_off);
                const uint32_t tiles = (uint32_t)((s.out_bytes + kPackTile - 1) / kPackTile);
                if (tiles) pack_copy_kernel<<<tiles, 256, 0, e->stream>>>(d_out, d_out_off, d_pack_off, (uint32_t)m, d_pack[k & 1]);
                PIPE_HIP(hipMemcpyAsync(meta + ML.sizes, d_sizes, m * 8, hipMemcpyDeviceToHost, e->stream));
                PIPE_HIP(hipMemcpyAsync(meta + ML.status, d_status, m * 4, hipMemcpyDeviceToHost, e->stream));
            } else {
                PIPE_HIP(hipMemcpyAsync(meta + ML.values, d_values, m * 4, hipMemcpyDeviceToHost, e->stream));
            }
            PIPE_HIP(hipGetLastError());
            PIPE_HIP(hipEventRecord(P->ev_k[k & 1], e->stream));
            {
                std::lock_guard<std::mutex> l(hm);
                launched = k + 1;
            }
            hcv.notify_all();
        }
        // ---- slice k - 1: its sizes are known, its packed bytes start their way back -------------
        if (k >= 1 && k - 1 < ns) {
            const size_t j = k - 1;
            const Slice &s = slices[j];
            const size_t m = s.i1 - s.i0;
            const uint8_t *meta = P->pin_meta[j & 1];
            PIPE_HIP(hipEventSynchronize(P->ev_k[j & 1]));
            if (!out_off) {
                memcpy(values + s.i0, meta + ML.values, m * 4);
            } else {
                memcpy(out_sizes + s.i0, meta + ML.sizes, m * 8);
                memcpy(status + s.i0, meta + ML.status, m * 4);
                // sizes are validated against the slots like src/batch_cuda.rs:123-137 does before touching the payload
                uint64_t total = 0;
                for (size_t i = s.i0; i < s.i1; i++) {
                    if (out_sizes[i] > out_off[i + 1] - out_off[i])
                        return fail(RSX_E_DEVICE, "device returned a size larger than its slot");
                    if (status[i] == 0) total += out_sizes[i];
                }
                packed[j] = total;
                direct[j] = total > P->pin_out_size;
                if (!direct[j]) {
                    if (total) PIPE_HIP(hipMemcpyAsync(P->pin_out[j & 1], d_pack[j & 1], total, hipMemcpyDeviceToHost, P->s_out));
                } else {
                    // a slice too large for the pinned buffer (one huge item): straight into the caller's slots
                    uint64_t at = 0;
                    for (size_t i = s.i0; i < s.i1; i++) {
                        if (status[i] != 0 || !out_sizes[i]) continue;
                        PIPE_HIP(hipMemcpyAsync(out + out_off[i], d_pack[j & 1] + at, out_sizes[i], hipMemcpyDeviceToHost, P->s_out));
                        at += out_sizes[i];
                    }
                }
                PIPE_HIP(hipEventRecord(P->ev_out[j & 1], P->s_out));
            }
        }
        // ---- slice k - 2: from the pinned buffer into the caller's slots ---------------------------
        if (k >= 2 && out_off) {
            const size_t i2 = k - 2;
            const Slice &s = slices[i2];
            PIPE_HIP(hipEventSynchronize(P->ev_out[i2 & 1]));
            if (!direct[i2] && packed[i2]) {
                // pieces of at most 1 MiB, so that a few large items keep every thread busy
                jobs.clear();
                const uint8_t *src = P->pin_out[i2 & 1];
                uint64_t at = 0;
                for (size_t i = s.i0; i < s.i1; i++) {
                    if (status[i] != 0 || !out_sizes[i]) continue;
                    for (uint64_t done = 0; done < out_sizes[i]; done += 1 << 20) {
                        const uint64_t len = out_sizes[i] - done < (1 << 20) ? out_sizes[i] - done : (1 << 20);
                        jobs.push_back(CopyJob{out + out_off[i] + done, src + at + done, (size_t)len});
                    }
                    at += out_sizes[i];
                }
                P->pool->run(jobs.size(), [&](size_t q) { memcpy(jobs[q].dst, jobs[q].src, jobs[q].len); });
            }
        }
    }
#undef PIPE_HIP
    helper.join();
    RSX_HIP(e, hipStreamSynchronize(e->stream));
    RSX_HIP(e, hipStreamSynchronize(P->s_out));
    if (op == 1) {
        e->d_last_parts = e->d_diag;
        e->d_last_fallbacks = e->d_diag + 3;
    }
    return RSX_OK;
}

// ---- engines bound to several GPUs ---------------------------------------------------------
namespace {

// Contiguous ranges of items, one per device, balanced by input bytes: cut r lies where the running sum
// comes closest to r / ndev of the total (an empty batch of bytes is cut by item count).
std::vector<size_t> shard_cuts(const uint64_t *in_off, size_t n, size_t ndev)
{
    std::vector<size_t> cuts(ndev + 1, 0);
    cuts[ndev] = n;
    const uint64_t total = n ? in_off[n] - in_off[0] : 0;
    for (size_t r = 1; r < ndev; r++) {
        size_t k;
        if (total == 0) {
            k = (size_t)((double)n * r / ndev + 0.5);
        } else {
            const double target = (double)total * r / ndev;
            size_t lo = 0, hi = n;   // first k with in_off[k] - in_off[0] >= target
            while (lo < hi) {
                const size_t mid = lo + (hi - lo) / 2;
                if ((double)(in_off[mid] - in_off[0]) < target) lo = mid + 1; else hi = mid;
            }
            k = lo;
            if (k > 0 && target - (double)(in_off[k - 1] - in_off[0]) <= (double)(in_off[k < n ? k : n] - in_off[0]) - target) k--;
        }
        if (k > n) k = n;
        cuts[r] = k < cuts[r - 1] ? cuts[r - 1] : k;
    }
    return cuts;
}

// runs f(engine of device r, first item, one past the last) for every range, one thread per device
template <class F>
int run_sharded(rsx_engine *e, const uint64_t *in_off, size_t n, F f)
{
    const size_t ndev = e->subs.size();
    const std::vector<size_t> cuts = shard_cuts(in_off, n, ndev);
    std::vector<int> rc(ndev, RSX_OK);
    std::vector<std::thread> th;
    for (size_t r = 1; r < ndev; r++)
        if (cuts[r + 1] > cuts[r]) th.emplace_back([&, r] { rc[r] = f(e->subs[r], cuts[r], cuts[r + 1]); });
    if (cuts[1] > cuts[0]) rc[0] = f(e->subs[0], cuts[0], cuts[1]);
    for (auto &t : th) t.join();
    for (size_t r = 0; r < ndev; r++)
        if (rc[r] != RSX_OK) {
            std::lock_guard<std::mutex> lk(e->mu);
            e->err = "device " + std::to_string(e->subs[r]->device) + ": " + e->subs[r]->err;
            return rc[r];
        }
    return RSX_OK;
}

}  // namespace

extern "C" {

int rsx_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return count;
}

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
        hipMalloc(reinterpret_cast<void **>(&e->d_diag), 256) != hipSuccess ||
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

int rsx_engine_create_multi(const int *devices, int ndev, rsx_engine **out)
{
    if (!out) return RSX_E_INVALID;
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return RSX_E_INVALID;
    if (ndev == 1) return rsx_engine_create(devices[0], nullptr, out);
    rsx_engine *e = new (std::nothrow) rsx_engine();
    if (!e) return RSX_E_NOMEM;
    e->device = -1;
    for (int r = 0; r < ndev; r++) {
        rsx_engine *sub = nullptr;
        const int rc = rsx_engine_create(devices[r], nullptr, &sub);
        if (rc != RSX_OK) {
            rsx_engine_destroy(e);
            return rc;
        }
        e->subs.push_back(sub);
    }
    *out = e;
    return RSX_OK;
}

int rsx_engine_num_devices(const rsx_engine *e) { return !e ? 0 : e->subs.empty() ? 1 : (int)e->subs.size(); }

void rsx_engine_destroy(rsx_engine *e)
{
    if (!e) return;
    if (!e->subs.empty() || e->device < 0) {
        for (rsx_engine *s : e->subs) rsx_engine_destroy(s);
        delete e;
        return;
    }
    DeviceGuard g(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    pipe_destroy(e);
    if (e->ws) (void)hipFree(e->ws);
    if (e->d_diag) (void)hipFree(e->d_diag);
    if (e->d_crc_tables) (void)hipFree(e->d_crc_tables);
    if (e->ev_begin) (void)hipEventDestroy(e->ev_begin);
    if (e->ev_end) (void)hipEventDestroy(e->ev_end);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int rsx_engine_set_option(rsx_engine *e, int option, int value)
{
    if (!e) return RSX_E_INVALID;
    if (option != RSX_OPT_VERIFY_FHCRC) return bad(e, "unknown option");
    for (rsx_engine *s : e->subs) (void)rsx_engine_set_option(s, option, value);
    std::lock_guard<std::mutex> lk(e->mu);
    e->verify_fhcrc = value != 0;
    return RSX_OK;
}

const char *rsx_engine_last_error(const rsx_engine *e) { return e ? e->err.c_str() : "null engine"; }

void *rsx_engine_stream(const rsx_engine *e) { return e && e->subs.empty() ? e->stream : nullptr; }

int rsx_engine_synchronize(rsx_engine *e)
{
    if (!e) return RSX_E_INVALID;
    for (rsx_engine *s : e->subs) {
        const int r = rsx_engine_synchronize(s);
        if (r != RSX_OK) return r;
    }
    if (!e->subs.empty()) return RSX_OK;
    DeviceGuard g(e->device);
    RSX_HIP(e, hipStreamSynchronize(e->stream));
    return RSX_OK;
}

float rsx_engine_last_kernel_ms(rsx_engine *e)
{
    if (!e) return -1.0f;
    if (!e->subs.empty()) {   // the slowest device
        float worst = -1.0f;
        for (rsx_engine *s : e->subs) {
            const float ms = rsx_engine_last_kernel_ms(s);
            if (ms > worst) worst = ms;
        }
        return worst;
    }
    if (!e->timed) return -1.0f;
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
    if (!e->subs.empty()) {   // summed over the devices
        bool any = false;
        for (rsx_engine *s : e->subs) {
            uint32_t w[3];
            if (last_part_words(s, w)) {
                any = true;
                for (int k = 0; k < 3; k++) v[k] += w[k];
            }
        }
        return any;
    }
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
    if (!e->subs.empty()) {
        uint32_t sum = 0;
        for (rsx_engine *s : e->subs) sum += rsx_engine_last_fallbacks(s);
        return sum;
    }
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

static int inflate_format(const rsx_engine *e, int format)
{
    return format == RSX_FORMAT_GZIP && e->verify_fhcrc ? format | kFormatVerifyFhcrc : format;
}

int rsx_batch_checksum_device(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                              const uint64_t *in_off, size_t n, uint64_t total_in,
                              uint64_t max_item, uint32_t *values)
{
    if (!e) return RSX_E_INVALID;
    if (!e->subs.empty()) return single_only(e);
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
    if (!e->subs.empty()) return single_only(e);
    if (!format_ok(format)) return bad(e, "bad format");
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    int r = ws_reserve(e, inflate_ws_bytes(n, total_in, total_out, e->inflate_split));
    if (r != RSX_OK) return r;
    return inflate_launch(e, inflate_format(e, format), in, in_off, n, total_in, total_out, max_out, out, out_off,
                          out_sizes, status, e->inflate_split);
}

int rsx_batch_decompress_device_noalloc(rsx_engine *e, int format, const uint8_t *in,
                                        const uint64_t *in_off, size_t n, uint64_t total_in,
                                        uint64_t total_out, uint64_t max_out, uint8_t *out,
                                        const uint64_t *out_off, uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!e->subs.empty()) return single_only(e);
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
    return inflate_launch(e, inflate_format(e, format), in, in_off, n, total_in, total_out, max_out, out, out_off,
                          out_sizes, status, false);
}

int rsx_batch_compress_device(rsx_engine *e, int format, int level, const uint8_t *in,
                              const uint64_t *in_off, size_t n, uint64_t total_in,
                              uint64_t max_item, uint8_t *out, const uint64_t *out_off,
                              uint64_t *out_sizes, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!e->subs.empty()) return single_only(e);
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

// one call with output slots, on however many devices the engine has
static int host_call(rsx_engine *e, int op, int format, int level, int kind, const uint8_t *in,
                     const uint64_t *in_off, size_t n, uint8_t *out, const uint64_t *out_off,
                     uint64_t *out_sizes, int32_t *status)
{
    if (e->subs.empty())
        return stage_and_run(e, op, op == 1 ? inflate_format(e, format) : format, level, kind, 0, in, in_off, n, out,
                             out_off, out_sizes, status, nullptr);
    for (size_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i]) return bad(e, "in_off not monotonic");
    return run_sharded(e, in_off, n, [&](rsx_engine *s, size_t a, size_t b) {
        return stage_and_run(s, op, op == 1 ? inflate_format(s, format) : format, level, kind, 0, in, in_off + a, b - a,
                             out, out_off + a, out_sizes + a, status + a, nullptr);
    });
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
    return host_call(e, 0, format, level, RSX_FLUSH_FINISH, in, in_off, n, out, out_off, out_sizes, status);
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
    return host_call(e, 0, RSX_FORMAT_DEFLATE, level, flush, in, in_off, n, out, out_off, out_sizes, status);
}

int rsx_batch_decompress(rsx_engine *e, int format, const uint8_t *in, const uint64_t *in_off,
                         size_t n, uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                         int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!format_ok(format)) return bad(e, "bad format");
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    return host_call(e, 1, format, 0, 0, in, in_off, n, out, out_off, out_sizes, status);
}

static int decompress_members_one(rsx_engine *e, const uint8_t *in, const uint64_t *in_off, size_t n,
                                  uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                                  int32_t *status)
{
    BatchShape si, so;
    if (!shape_of(in_off, n, &si)) return bad(e, "in_off not monotonic");
    if (!shape_of(out_off, n, &so)) return bad(e, "out_off not monotonic");
    if ((!in && si.total) || (!out && so.total)) return bad(e, "null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    const int format = inflate_format(e, RSX_FORMAT_GZIP);
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
    // serial launches while every other item waits: items still unfinished after RSX_MAX_GZIP_MEMBERS members are
    // reported RSX_BAD_DATA (gzip sets no limit; pigz and concatenated archives stay far below this one).
    size_t rounds_done = 0;
    auto rounds = [&]() -> int {
        while (!active.empty()) {
            if (rounds_done++ == RSX_MAX_GZIP_MEMBERS) {
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
            const int rr = inflate_launch(e, format, d_in, d_lo, m, si.total, so.total, max_out, d_out, d_olo, d_sizes,
                                          d_status, e->inflate_split, &sl);
            if (rr != RSX_OK) return rr;
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

int rsx_batch_decompress_members(rsx_engine *e, const uint8_t *in, const uint64_t *in_off, size_t n,
                                 uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                                 int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (n == 0) return RSX_OK;
    if (!in_off || !out_off || !out_sizes || !status) return bad(e, "null pointer");
    if (e->subs.empty()) return decompress_members_one(e, in, in_off, n, out, out_off, out_sizes, status);
    for (size_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i]) return bad(e, "in_off not monotonic");
    return run_sharded(e, in_off, n, [&](rsx_engine *s, size_t a, size_t b) {
        return decompress_members_one(s, in, in_off + a, b - a, out, out_off + a, out_sizes + a, status + a);
    });
}

int rsx_inflate_stream_step(rsx_engine *e, const uint8_t *in, size_t in_len, uint32_t in_bit,
                            const uint8_t *history, size_t history_len, uint8_t *out, size_t out_cap,
                            uint64_t *in_next_bit, size_t *out_produced, int32_t *status)
{
    if (!e) return RSX_E_INVALID;
    if (!e->subs.empty()) e = e->subs[0];   // (one stream is one workgroup's work: the first GPU)
    if (!in_next_bit || !out_produced || !status || (!in && in_len) || (!history && history_len) || (!out && out_cap))
        return bad(e, "null pointer");
    if (in_bit > 7) return bad(e, "in_bit must be 0..7");
    if (history_len > 32768) {   // a match reaches back 32 KiB at most
        history += history_len - 32768;
        history_len = 32768;
    }
    if ((uint64_t)out_cap + history_len >= (1ull << 32) - (1ull << 16) || (uint64_t)in_len >= (1ull << 32))
        return bad(e, "step too large (input and output room must be < 4 GiB)");
    std::lock_guard<std::mutex> lk(e->mu);
    DeviceGuard g(e->device);
    const uint64_t slot = (uint64_t)history_len + out_cap;
    int r = ws_reserve(e, align_up(in_len + 64, 256) + align_up(slot + 64, 256) + 1024);
    if (r != RSX_OK) return r;
    uint8_t *d_in = static_cast<uint8_t *>(ws_take(e, in_len + 64));
    uint8_t *d_out = static_cast<uint8_t *>(ws_take(e, slot + 64));
    uint64_t *ctl = static_cast<uint64_t *>(ws_take(e, 10 * sizeof(uint64_t)));
    if (!d_in || !d_out || !ctl) {
        e->err = "workspace carve failed";
        return RSX_E_NOMEM;
    }
    uint64_t h[10] = {0, in_len, 0, slot, history_len, 0, 0, 0, in_bit, 0};
    if (in_len) RSX_HIP(e, hipMemcpyAsync(d_in, in, in_len, hipMemcpyHostToDevice, e->stream));
    if (history_len) RSX_HIP(e, hipMemcpyAsync(d_out, history, history_len, hipMemcpyHostToDevice, e->stream));
    RSX_HIP(e, hipMemcpyAsync(ctl, h, sizeof h, hipMemcpyHostToDevice, e->stream));
    RSX_HIP(e, hipStreamSynchronize(e->stream));   // (h lives on this stack)
    r = inflate_resume_launch(e, d_in, in_len, d_out, ctl);
    if (r != RSX_OK) return r;
    RSX_HIP(e, hipMemcpyAsync(h, ctl, sizeof h, hipMemcpyDeviceToHost, e->stream));
    RSX_HIP(e, hipStreamSynchronize(e->stream));
    const uint64_t produced = h[6], next_bit = h[5];
    const int32_t st = (int32_t)(uint32_t)h[9];
    if (produced > out_cap || next_bit > (uint64_t)in_len * 8) {
        e->err = "device returned a position outside the step's buffers";
        return RSX_E_DEVICE;
    }
    if (produced) {
        RSX_HIP(e, hipMemcpyAsync(out, d_out + history_len, produced, hipMemcpyDeviceToHost, e->stream));
        RSX_HIP(e, hipStreamSynchronize(e->stream));
    }
    *in_next_bit = st == RSX_INTERNAL_ERROR ? in_bit : next_bit;
    *out_produced = (size_t)produced;
    *status = st;
    return RSX_OK;
}

int rsx_batch_checksum(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                       const uint64_t *in_off, size_t n, uint32_t *values)
{
    if (!e) return RSX_E_INVALID;
    if (kind != RSX_CHECKSUM_CRC32 && kind != RSX_CHECKSUM_ADLER32) return bad(e, "bad checksum kind");
    if (n == 0) return RSX_OK;
    if (!in_off || !values) return bad(e, "null pointer");
    if (e->subs.empty())
        return stage_and_run(e, 2, 0, 0, kind, init, in, in_off, n, nullptr, nullptr, nullptr, nullptr, values);
    for (size_t i = 0; i < n; i++)
        if (in_off[i + 1] < in_off[i]) return bad(e, "in_off not monotonic");
    return run_sharded(e, in_off, n, [&](rsx_engine *s, size_t a, size_t b) {
        return stage_and_run(s, 2, 0, 0, kind, init, in, in_off + a, b - a, nullptr, nullptr, nullptr, nullptr, values + a);
    });
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
