// checksum.hip -- batch CRC-32 and Adler-32 on gfx950.
//
// Replaces the CPU checksum passes of the reference: crc32()
// (src/crc32/mod.rs:332-365) and adler32() (src/adler32/mod.rs:115-152), which
// the zlib / gzip wrappers run over the whole uncompressed buffer
// (src/compress/mod.rs:2288, :2339; src/decompress/mod.rs:1106, :1204).
//
// Data layout: items are slices [in_off[i], in_off[i+1]) of one flat device
// buffer.  Each item is cut into 32 KiB segments; one 256-thread workgroup
// reduces one segment with 16-byte coalesced loads, a second small kernel
// folds the per-segment partials of each item in order.
//
// CRC-32 is linear over GF(2): the raw register after A||B equals
// reg(A) * x^(8|B|) xor reg0(B) (mod P, reflected 0xEDB88320).  Inside a
// segment lane l reduces the 16-byte units l, l+256, ... with a strided Horner
// step s = update16(s * x^(128*255), unit); the four "multiply by a constant"
// and four slice-by-4 tables live in LDS.  Adler-32 is reduced as the two sums
// S1 = sum d_i and sum (L - i) d_i, which compose exactly under mod 65521.
//
// Roofline: HBM-bound, algorithmic bytes = item length (each byte read once).
#include "engine.h"

#include <vector>

namespace {

constexpr int kThreads = 256;
constexpr uint32_t kSeg = 32768;  // bytes per partial result
#ifndef RSX_CRC_SEGS_PER_WG
#define RSX_CRC_SEGS_PER_WG 4
#endif
constexpr uint32_t kSegsPerWg = RSX_CRC_SEGS_PER_WG;  // segments a workgroup reduces in turn (the tables are loaded once)
constexpr uint32_t kPoly = 0xEDB88320u;
constexpr uint32_t kAdlerMod = 65521u;

// table offsets (in u32) inside rsx_engine::d_crc_tables
#ifndef RSX_CRC_GROUP
#define RSX_CRC_GROUP 4
#endif
constexpr uint32_t kGroup = RSX_CRC_GROUP;   // consecutive 16-byte units a lane takes between two multiplications
constexpr int kTabSlice = 0;     // 4 x 256: slice-by-4 tables
constexpr int kTabMulC = 1024;   // 4 x 256: multiply a state byte by x^(128*kGroup*(kThreads-1))
constexpr int kTabPow = 2048;    // kGroup * 256: x^(128*j)
constexpr int kTabXSeg = kTabPow + kGroup * 256;   // x^(8*kSeg)
constexpr int kTabWords = kTabXSeg + 1;

// ---- GF(2)[x] mod P helpers in the reflected representation (x^0 = 1<<31) ----

__host__ __device__ inline uint32_t multmodp(uint32_t a, uint32_t b)
{
    uint32_t p = 0;
    for (int i = 31; i >= 0; i--) {  // fixed trip count: a == 0 must terminate
        p ^= b & (0u - ((a >> i) & 1));
        b = (b >> 1) ^ (kPoly & (0u - (b & 1)));
    }
    return p;
}

// x^(8*nbytes) mod P by square and multiply
__host__ __device__ inline uint32_t xpow8(uint64_t nbytes)
{
    uint32_t r = 1u << 31;          // x^0
    uint32_t sq = 1u << (31 - 8);   // x^8
    while (nbytes) {
        if (nbytes & 1) r = multmodp(r, sq);
        sq = multmodp(sq, sq);
        nbytes >>= 1;
    }
    return r;
}

struct SegInfo {
    const uint8_t *base;  // item start
    uint64_t len;         // item length
    uint32_t seg;         // segment index inside the item
    bool valid;
};

// Maps a workgroup to (item, segment).  Uniform mode: every item owns
// segs_per_item slots.  General mode: binary search in the exclusive prefix
// seg_start[0..n].
__device__ inline uint64_t item_len(const uint64_t *starts, const uint64_t *lens, uint32_t i)
{
    return lens ? lens[i] : starts[i + 1] - starts[i];
}

__device__ inline SegInfo locate(const uint8_t *in, const uint64_t *starts, const uint64_t *lens,
                                 uint32_t n, const uint32_t *seg_start, uint32_t segs_per_item,
                                 uint32_t slot, uint32_t *item_ret)
{
    SegInfo s{};
    uint32_t item, seg;
    if (seg_start == nullptr) {
        item = slot / segs_per_item;
        seg = slot % segs_per_item;
        if (item >= n) return s;
    } else {
        uint32_t g = slot;
        if (g >= seg_start[n]) return s;
        uint32_t lo = 0, hi = n;  // find item with seg_start[item] <= g < seg_start[item+1]
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (seg_start[mid] <= g) lo = mid; else hi = mid;
        }
        item = lo;
        seg = g - seg_start[item];
    }
    s.base = in + starts[item];
    s.len = item_len(starts, lens, item);
    s.seg = seg;
    s.valid = (uint64_t)seg * kSeg < s.len;
    *item_ret = item;
    return s;
}

__device__ inline uint32_t wg_reduce_xor(uint32_t v, uint32_t *lds)
{
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o);
    WG_SYNC();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    WG_SYNC();
    return lds[0] ^ lds[1] ^ lds[2] ^ lds[3];
}

__device__ inline uint64_t wg_reduce_add(uint64_t v, uint64_t *lds)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    WG_SYNC();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    WG_SYNC();
    return lds[0] + lds[1] + lds[2] + lds[3];
}

// One workgroup reduces kSegsPerWg segments (slots), one after the other.  partial[slot] receives
//   CRC-32 : {raw register of the segment from a zero start, 0}
//   Adler-32: {S1 mod 65521, sum (L - i) d_i mod 65521}
template <int KIND>
__global__ __launch_bounds__(kThreads) void checksum_partial_kernel(
    const uint8_t *__restrict__ in, const uint64_t *__restrict__ starts,
    const uint64_t *__restrict__ lens, uint32_t n, const uint32_t *__restrict__ seg_start,
    uint32_t segs_per_item, const uint32_t *__restrict__ tables, uint2 *__restrict__ partial)
{
    __shared__ uint32_t tab[2048];
    __shared__ uint64_t red[8];
    const int tid = threadIdx.x;
    if (KIND == RSX_CHECKSUM_CRC32) {
        for (int i = tid; i < 2048; i += kThreads) tab[i] = tables[i];
    }
    WG_SYNC();
    const uint32_t slot_end = (blockIdx.x + 1) * kSegsPerWg;
    for (uint32_t slot = blockIdx.x * kSegsPerWg, run = 1; slot < slot_end; slot += run) {
    uint32_t item;
    run = 1;
    SegInfo s = locate(in, starts, lens, n, seg_start, segs_per_item, slot, &item);
    if (!s.valid) continue;  // slot past the end of a short item or of the batch: never read by the fold (uniform)
    // CRC-32: the workgroup's slots that hold consecutive segments of this item are reduced in one go (one
    // final multiplication and one reduction over the lanes for all of them); the result goes to the last
    // of the slots and zeros to the others, which the fold turns into the same value.
    if (KIND == RSX_CHECKSUM_CRC32) {
        while (slot + run < slot_end) {
            uint32_t item2;
            const SegInfo s2 = locate(in, starts, lens, n, seg_start, segs_per_item, slot + run, &item2);
            if (!s2.valid || item2 != item || s2.seg != s.seg + run) break;
            run++;
        }
    }
    const uint64_t seg_begin = (uint64_t)s.seg * kSeg;
    const uint32_t L = (uint32_t)((s.len - seg_begin) < (uint64_t)run * kSeg ? (s.len - seg_begin) : (uint64_t)run * kSeg);
    const uint8_t *p = s.base + seg_begin;
    // 16-byte unit grid over the global address space
    const uintptr_t addr = (uintptr_t)p;
    const uint32_t head = (uint32_t)(addr & 15);          // bytes of unit 0 before the data
    const uint8_t *grid = p - head;                        // 16-byte aligned
    const uint32_t span = head + L;                        // bytes from grid start to data end
    const uint32_t full_units = span >> 4;                 // units that end inside the data
    const uint32_t tail = span & 15;                       // bytes after the last full unit

    if (KIND == RSX_CHECKSUM_CRC32) {
        // lane l takes the groups l, l + kThreads, ... of kGroup consecutive units: one multiplication by
        // x^(128 * kGroup * (kThreads - 1)) carries its state from the end of one group to the start of the next
        uint32_t st = 0;
        int last_u = -1;
        for (uint32_t g0 = tid * kGroup; g0 < full_units; g0 += kThreads * kGroup) {
            uint4 v[kGroup];
#pragma unroll
            for (uint32_t j = 0; j < kGroup; j++)   // (every load is issued before the first one is needed)
                v[j] = g0 + j < full_units ? *reinterpret_cast<const uint4 *>(grid + 16 * (size_t)(g0 + j))
                                           : make_uint4(0, 0, 0, 0);
            if (g0 == 0 && head) {  // zero the bytes in front of the data (leading zeros are free)
                uint32_t w[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
                for (uint32_t b = 0; b < head; b++) w[b >> 2] &= ~(0xFFu << (8 * (b & 3)));
                v[0] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            st = tab[kTabMulC + (st & 0xFF)] ^ tab[kTabMulC + 256 + ((st >> 8) & 0xFF)] ^
                 tab[kTabMulC + 512 + ((st >> 16) & 0xFF)] ^ tab[kTabMulC + 768 + (st >> 24)];
#pragma unroll
            for (uint32_t j = 0; j < kGroup; j++) {
                if (g0 + j < full_units) {
                    const uint32_t w4[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        uint32_t t = st ^ w4[k];
                        st = tab[768 + (t & 0xFF)] ^ tab[512 + ((t >> 8) & 0xFF)] ^
                             tab[256 + ((t >> 16) & 0xFF)] ^ tab[t >> 24];
                    }
                    last_u = (int)(g0 + j);
                }
            }
        }
        if (last_u >= 0) {
            uint32_t behind = full_units - 1 - (uint32_t)last_u;  // units after mine, < kGroup * kThreads
            st = multmodp(st, tables[kTabPow + behind]);
        }
        uint32_t crc = wg_reduce_xor(st, reinterpret_cast<uint32_t *>(red));
        if (tid == 0) {
            // the last partial unit, byte by byte; when there is no full unit the
            // data starts inside this unit
            uint32_t from = full_units ? 0 : head;
            const uint8_t *t = grid + 16 * (size_t)full_units;
            for (uint32_t b = from; b < tail; b++) crc = (crc >> 8) ^ tab[(crc ^ t[b]) & 0xFF];
            for (uint32_t k = 0; k + 1 < run; k++) partial[slot + k] = make_uint2(0, 0);
            partial[slot + run - 1] = make_uint2(crc, 0);
        }
    } else {
        // weights: byte at grid offset g has data index i = g - head and weight (L - i)
        uint32_t s1 = 0, wsum = 0;  // wsum = sum g * d  (g < 32784, <= 144 bytes per lane)
        const uint32_t units = (span + 15) >> 4;
        for (uint32_t u = tid; u < units; u += kThreads) {
            uint4 v = *reinterpret_cast<const uint4 *>(grid + 16 * (size_t)u);
            uint32_t w4[4] = {v.x, v.y, v.z, v.w};
            uint32_t g0 = u * 16;
#pragma unroll
            for (int b = 0; b < 16; b++) {
                uint32_t g = g0 + b;
                uint32_t d = (w4[b >> 2] >> (8 * (b & 3))) & 0xFF;
                d = (g >= head && g < span) ? d : 0;
                s1 += d;
                wsum += g * d;
            }
        }
        uint64_t S1 = wg_reduce_add(s1, red);
        uint64_t W = wg_reduce_add(wsum, red);
        if (tid == 0) {
            // sum (L - i) d_i = (L + head) * S1 - W
            uint64_t s2 = (uint64_t)(L + head) * S1 - W;
            partial[slot] = make_uint2((uint32_t)(S1 % kAdlerMod), (uint32_t)(s2 % kAdlerMod));
        }
    }
    }
}

// One thread per item folds that item's partials in order.
template <int KIND>
__global__ void checksum_combine_kernel(const uint64_t *__restrict__ starts,
                                        const uint64_t *__restrict__ lens, uint32_t n,
                                        const uint32_t *__restrict__ seg_start,
                                        uint32_t segs_per_item, const uint32_t *__restrict__ tables,
                                        const uint2 *__restrict__ partial, uint64_t num_slots, uint32_t init,
                                        uint32_t *__restrict__ values)
{
    uint32_t item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n) return;
    uint64_t len = item_len(starts, lens, item);
    uint32_t nseg = (uint32_t)((len + kSeg - 1) / kSeg);
    size_t first = seg_start ? seg_start[item] : (size_t)item * segs_per_item;
    // the items are longer than the caller's total said: there are no partials for this one
    if (first + nseg > num_slots || (!seg_start && nseg > segs_per_item)) {
        values[item] = 0;
        return;
    }
    if (KIND == RSX_CHECKSUM_CRC32) {
        uint32_t reg = ~init;
        const uint32_t xseg = tables[kTabXSeg];
        for (uint32_t k = 0; k < nseg; k++) {
            uint64_t L = (k + 1 == nseg) ? len - (uint64_t)k * kSeg : kSeg;
            uint32_t xp = (L == kSeg) ? xseg : xpow8(L);
            reg = multmodp(reg, xp) ^ partial[first + k].x;
        }
        values[item] = ~reg;
    } else {
        uint64_t a = init & 0xFFFF, b = init >> 16;
        for (uint32_t k = 0; k < nseg; k++) {
            uint64_t L = (k + 1 == nseg) ? len - (uint64_t)k * kSeg : kSeg;
            uint2 p = partial[first + k];
            b = (b + (L % kAdlerMod) * a + p.y) % kAdlerMod;
            a = (a + p.x) % kAdlerMod;
        }
        values[item] = (uint32_t)((b << 16) | a);
    }
}

// Exclusive prefix sum of per-item segment counts, single workgroup.
__global__ __launch_bounds__(1024) void seg_scan_kernel(const uint64_t *__restrict__ starts,
                                                         const uint64_t *__restrict__ lens,
                                                         uint32_t n, uint32_t *__restrict__ seg_start)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    WG_SYNC();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t c = 0;
        if (i < n) c = (uint32_t)((item_len(starts, lens, i) + kSeg - 1) / kSeg);
        uint32_t x = c;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        WG_SYNC();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < n) seg_start[i] = before + x - c;
        WG_SYNC();
        if (threadIdx.x == 1023) carry = before + x;
        WG_SYNC();
    }
    if (threadIdx.x == 0) seg_start[n] = carry;
}

}  // namespace

int checksum_tables_create(rsx_engine *e)
{
    std::vector<uint32_t> t(kTabWords);
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++) c = (c >> 1) ^ (kPoly & (0u - (c & 1)));
        t[kTabSlice + i] = c;
    }
    for (int k = 1; k < 4; k++)
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t prev = t[kTabSlice + (k - 1) * 256 + i];
            t[kTabSlice + k * 256 + i] = (prev >> 8) ^ t[kTabSlice + (prev & 0xFF)];
        }
    const uint32_t C = xpow8(16ull * kGroup * (kThreads - 1));
    for (int k = 0; k < 4; k++)
        for (uint32_t i = 0; i < 256; i++) t[kTabMulC + k * 256 + i] = multmodp(i << (8 * k), C);
    for (uint32_t j = 0; j < kGroup * 256; j++) t[kTabPow + j] = xpow8(16ull * j);
    t[kTabXSeg] = xpow8(kSeg);
    RSX_HIP(e, hipMalloc(&e->d_crc_tables, kTabWords * sizeof(uint32_t)));
    RSX_HIP(e, hipMemcpy(e->d_crc_tables, t.data(), kTabWords * sizeof(uint32_t),
                         hipMemcpyHostToDevice));
    return RSX_OK;
}

static inline uint64_t seg_upper_bound(size_t n, uint64_t total_in) { return total_in / kSeg + n; }

size_t checksum_ws_bytes(size_t n, uint64_t total_in)
{
    // partials for either launch mode (uniform mode is only chosen when it needs
    // at most twice the general bound) + the segment prefix
    return align_up(2 * seg_upper_bound(n, total_in) * sizeof(uint2) + 256, 256) +
           align_up((n + 1) * sizeof(uint32_t), 256);
}

int checksum_launch(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                    const uint64_t *in_off, const uint64_t *lens, size_t n, uint64_t total_in,
                    uint64_t max_item, uint32_t *values, bool trusted_max)
{
    if (n == 0) return RSX_OK;
    if (n > 0x7FFFFFFFull) {
        e->err = "batch too large";
        return RSX_E_INVALID;
    }
    const uint64_t bound = seg_upper_bound(n, total_in);
    const uint64_t segs_per_item = (max_item + kSeg - 1) / kSeg;
    // Uniform mode gives every item ceil(max_item / kSeg) slots without looking at the lengths: an
    // understated max_item would make the fold read a neighbour's partials, so it is only taken when
    // this library computed max_item itself.
    const bool uniform = trusted_max && segs_per_item * n <= 2 * bound && segs_per_item > 0;
    uint64_t grid = ((uniform ? segs_per_item * n : bound) + kSegsPerWg - 1) / kSegsPerWg;
    if (grid == 0) grid = 1;
    if (grid > 0x7FFFFFFFull) {
        e->err = "checksum launch too large";
        return RSX_E_INVALID;
    }
    uint2 *partial = static_cast<uint2 *>(ws_take(e, 2 * bound * sizeof(uint2) + 256));
    uint32_t *seg_start = static_cast<uint32_t *>(ws_take(e, (n + 1) * sizeof(uint32_t)));
    if (!partial || !seg_start) {
        e->err = "workspace exhausted";
        return RSX_E_NOMEM;
    }
    const uint32_t *seg_arg = nullptr;
    if (!uniform) {
        seg_scan_kernel<<<1, 1024, 0, e->stream>>>(in_off, lens, (uint32_t)n, seg_start);
        seg_arg = seg_start;
    }
    const uint32_t spi = (uint32_t)segs_per_item;
    const uint32_t cgrid = (uint32_t)((n + 255) / 256);
    if (kind == RSX_CHECKSUM_CRC32) {
        checksum_partial_kernel<RSX_CHECKSUM_CRC32><<<(uint32_t)grid, kThreads, 0, e->stream>>>(
            in, in_off, lens, (uint32_t)n, seg_arg, spi, e->d_crc_tables, partial);
        checksum_combine_kernel<RSX_CHECKSUM_CRC32><<<cgrid, 256, 0, e->stream>>>(
            in_off, lens, (uint32_t)n, seg_arg, spi, e->d_crc_tables, partial, grid * kSegsPerWg, init, values);
    } else {
        checksum_partial_kernel<RSX_CHECKSUM_ADLER32><<<(uint32_t)grid, kThreads, 0, e->stream>>>(
            in, in_off, lens, (uint32_t)n, seg_arg, spi, e->d_crc_tables, partial);
        checksum_combine_kernel<RSX_CHECKSUM_ADLER32><<<cgrid, 256, 0, e->stream>>>(
            in_off, lens, (uint32_t)n, seg_arg, spi, e->d_crc_tables, partial, grid * kSegsPerWg, init, values);
    }
    RSX_HIP(e, hipGetLastError());
    return RSX_OK;
}
