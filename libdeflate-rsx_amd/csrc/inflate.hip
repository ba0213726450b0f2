// inflate.hip -- batch raw-DEFLATE / zlib / gzip decompression on gfx950.
//
// GPU counterpart of BatchDecompressor::decompress_batch (src/batch.rs:74-101)
// and of the per-item decoder it drives: block state machine
// (src/decompress/mod.rs:213-353), dynamic header (:403-507), decode-table
// build (:1324-1495), Huffman block body + LZ copy (:509-1072), zlib / gzip
// framing (:1074-1240).  Result codes are the reference's DecompressResult.
//
// One 256-thread workgroup decodes one stream in rounds:
//   1. stage   a 4 KiB window of compressed bytes into LDS with 16-byte
//              coalesced loads;
//   2. decode  Huffman symbols from that window against decode tables held in
//              LDS: literals go straight into a 32 KiB LDS ring that mirrors
//              the tail of the output, matches are queued as (dst, len, dist);
//   3. copy    all lanes resolve the queued matches in parallel -- a match may
//              run once every byte below its source end is final; sources older
//              than the ring are read back from the already written output;
//   4. flush   the round's bytes leave the ring as 16-byte coalesced stores.
// HBM traffic is therefore one read of the compressed bytes and one write of
// the output; the history window is served from LDS (and L2 for far matches).
//
// Roofline: HBM-bound by design, algorithmic bytes per stream =
// compressed_len + uncompressed_len.
#include "engine.h"

namespace {

constexpr int kT = 256;                       // threads per stream
constexpr int kRingBits = 15;
constexpr uint32_t kRing = 1u << kRingBits;   // LDS mirror of the newest output bytes
constexpr uint32_t kRingMask = kRing - 1;
constexpr uint32_t kWinBytes = 4096;          // staged compressed bytes per round
constexpr uint32_t kWinWords = kWinBytes / 4;
// a token may start only while two dwords past its last fetch are still staged:
// litlen fetch at b, offset fetch at most 20 bits later, each reads words w, w+1
constexpr uint32_t kWinLimitBits = (kWinWords - 2) * 32 - 20;
constexpr uint32_t kRoundOut = 8192;          // output bytes per round (<= kRing)
constexpr uint32_t kMaxMatches = 768;         // match queue entries per round

constexpr int kLitlenBits = 10;
constexpr int kLitlenEnough = 1334;           // enough(288, 10, 15)
constexpr int kOffsetBits = 8;
constexpr int kOffsetEnough = 402;            // enough(32, 8, 15)
constexpr int kPrecodeBits = 7;
constexpr int kPrecodeEnough = 128;

// decode-table entry flags (same meaning as src/decompress/tables.rs:32-35)
constexpr uint32_t kLiteral = 0x80000000u;
constexpr uint32_t kExceptional = 0x00008000u;
constexpr uint32_t kSubtable = 0x00004000u;
constexpr uint32_t kEndOfBlock = 0x00002000u;

enum { KIND_PRECODE = 0, KIND_LITLEN = 1, KIND_OFFSET = 2 };

__constant__ uint16_t c_len_base[31] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,
                                        17, 19, 23, 27, 31, 35, 43, 51, 59,  67,  83,
                                        99, 115, 131, 163, 195, 227, 258, 258, 258};
__constant__ uint8_t c_len_extra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                        3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
__constant__ uint16_t c_off_base[32] = {1,    2,    3,    4,    5,    7,     9,     13,
                                        17,   25,   33,   49,   65,   97,    129,   193,
                                        257,  385,  513,  769,  1025, 1537,  2049,  3073,
                                        4097, 6145, 8193, 12289, 16385, 24577, 24577, 24577};
__constant__ uint8_t c_off_extra[32] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,  4,  5,  5,  6,  6,
                                        7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 13, 13};
__constant__ uint8_t c_precode_perm[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                           11, 4,  12, 3, 13, 2, 14, 1, 15};

// per-symbol decode results, src/decompress/tables.rs:10-30, :55-379
__device__ inline uint32_t decode_result(int kind, uint32_t sym)
{
    if (kind == KIND_PRECODE) return sym << 16;
    if (kind == KIND_LITLEN) {
        if (sym < 256) return kLiteral | (sym << 16);
        if (sym == 256) return kExceptional | kEndOfBlock;
        return ((uint32_t)c_len_base[sym - 257] << 16) | c_len_extra[sym - 257];
    }
    return ((uint32_t)c_off_base[sym] << 16) | c_off_extra[sym];
}

__device__ inline uint32_t make_entry(int kind, uint32_t sym, uint32_t len)
{
    return decode_result(kind, sym) + (len << 8) + len;
}

// Everything a workgroup keeps in LDS.
struct __align__(16) StreamLds {
    uint8_t ring[kRing];
    uint32_t win[kWinWords + 2];
    uint32_t litlen[kLitlenEnough];
    uint32_t offset[kOffsetEnough];
    uint32_t precode[kPrecodeEnough];
    uint2 matches[kMaxMatches];   // x = (dst - round_start) | len << 16, y = dist
    uint16_t sorted_syms[288];
    uint8_t lens[288 + 32 + 140];
    uint8_t precode_lens[20];
    uint32_t first_open[kT / 64];
    // control block written by thread 0, read by all after a barrier
    uint32_t litlen_tablebits;
    uint32_t num_matches;
    uint32_t round_start_lo, round_end_lo;  // output positions (streams are < 4 GiB)
    int32_t status;       // -1 while running, else a DecompressResult
    int32_t in_block;     // 1 while inside a Huffman block body
    int32_t final_block;
    int32_t stored_len;   // >= 0: a stored block of that many bytes is pending
    uint64_t bitpos;      // next unread bit, relative to the start of the DEFLATE body
};

// Serial decode-table build; restates src/decompress/mod.rs:1324-1495 (called by
// one thread).  Returns false when the code-length set is rejected.
__device__ __noinline__ bool build_decode_table(uint32_t *table, const uint8_t *lens, uint32_t num_syms,
                                   int kind, uint32_t table_bits, uint32_t max_codeword_len,
                                   uint16_t *sorted_syms, uint32_t *table_bits_ret)
{
    uint32_t len_counts[16];
    uint32_t offsets[16];
    for (int i = 0; i < 16; i++) len_counts[i] = 0;
    for (uint32_t s = 0; s < num_syms; s++) len_counts[lens[s]]++;
    while (max_codeword_len > 1 && len_counts[max_codeword_len] == 0) max_codeword_len--;
    if (table_bits_ret) {
        if (table_bits > max_codeword_len) table_bits = max_codeword_len;
        *table_bits_ret = table_bits;
    }
    offsets[0] = 0;
    offsets[1] = len_counts[0];
    uint32_t codespace_used = 0;
    for (uint32_t len = 1; len < max_codeword_len; len++) {
        offsets[len + 1] = offsets[len] + len_counts[len];
        codespace_used = (codespace_used << 1) + len_counts[len];
    }
    codespace_used = (codespace_used << 1) + len_counts[max_codeword_len];
    for (uint32_t s = 0; s < num_syms; s++) sorted_syms[offsets[lens[s]]++] = (uint16_t)s;

    if (codespace_used > (1u << max_codeword_len)) return false;
    if (codespace_used < (1u << max_codeword_len)) {
        uint32_t s;
        if (codespace_used == 0) {
            s = 0;
        } else {
            if (codespace_used != (1u << (max_codeword_len - 1)) || len_counts[1] != 1) return false;
            s = sorted_syms[offsets[0]];
        }
        uint32_t e = make_entry(kind, s, 1);
        for (uint32_t i = 0; i < (1u << table_bits); i++) table[i] = e;
        return true;
    }

    const uint16_t *sp = sorted_syms + offsets[0];
    uint32_t codeword = 0;
    uint32_t len = 1;
    while (len_counts[len] == 0) len++;
    uint32_t cur_table_end = 1u << len;
    while (len <= table_bits) {
        uint32_t count = len_counts[len];
        while (count--) {
            table[codeword] = make_entry(kind, *sp++, len);
            if (codeword == cur_table_end - 1) {
                for (; len < table_bits; len++) {
                    for (uint32_t i = 0; i < cur_table_end; i++) table[cur_table_end + i] = table[i];
                    cur_table_end <<= 1;
                }
                return true;
            }
            uint32_t bit = 1u << (31 - __clz(codeword ^ (cur_table_end - 1)));
            codeword &= bit - 1;
            codeword |= bit;
        }
        do {
            if (++len <= table_bits) {
                for (uint32_t i = 0; i < cur_table_end; i++) table[cur_table_end + i] = table[i];
                cur_table_end <<= 1;
            }
        } while (len <= table_bits && len_counts[len] == 0);
        if (len > table_bits)
            while (len_counts[len] == 0) len++;
    }

    cur_table_end = 1u << table_bits;
    uint32_t subtable_prefix = ~0u;
    uint32_t subtable_start = 0;
    for (;;) {
        if ((codeword & ((1u << table_bits) - 1)) != subtable_prefix) {
            subtable_prefix = codeword & ((1u << table_bits) - 1);
            subtable_start = cur_table_end;
            uint32_t subtable_bits = len - table_bits;
            uint32_t sub_used = len_counts[len];
            while (sub_used < (1u << subtable_bits)) {
                subtable_bits++;
                sub_used = (sub_used << 1) +
                           (table_bits + subtable_bits <= 15 ? len_counts[table_bits + subtable_bits] : 0);
            }
            cur_table_end = subtable_start + (1u << subtable_bits);
            table[subtable_prefix] = (subtable_start << 16) | kExceptional | kSubtable |
                                     (subtable_bits << 8) | table_bits;
        }
        uint32_t e = make_entry(kind, *sp++, len - table_bits);
        uint32_t stride = 1u << (len - table_bits);
        for (uint32_t i = subtable_start + (codeword >> table_bits); i < cur_table_end; i += stride)
            table[i] = e;
        if (codeword == (1u << len) - 1) return true;
        uint32_t bit = 1u << (31 - __clz(codeword ^ ((1u << len) - 1)));
        codeword &= bit - 1;
        codeword |= bit;
        len_counts[len]--;
        while (len_counts[len] == 0) len++;
    }
}

// 32 valid bits starting at window bit offset b
__device__ inline uint32_t fetch_bits(const uint32_t *win, uint32_t b)
{
    uint32_t w = b >> 5;
    uint64_t v = ((uint64_t)win[w + 1] << 32) | win[w];
    return (uint32_t)(v >> (b & 31));
}

struct StreamCtx {
    const uint8_t *in_alloc;      // start of the whole flat input allocation
    uint64_t in_alloc_len;
    const uint8_t *body;          // first byte of the DEFLATE body
    uint64_t body_len;
    uint8_t *out;                 // output slot
    uint64_t out_cap;
    uint32_t shift;               // (uintptr_t)out & 15: ring index = (pos + shift) & mask
};

// Stage kWinBytes of compressed input around the current bit position.  Returns
// (through *win_bit0) the body-relative bit position of window bit 0.
__device__ inline void stage_window(StreamLds &L, const StreamCtx &c, int64_t *win_bit0)
{
    const uint64_t byte_pos = L.bitpos >> 3;
    const uint8_t *cur = c.body + byte_pos;
    const uint8_t *wbase = cur - ((uintptr_t)cur & 15);
    *win_bit0 = ((int64_t)(wbase - c.body)) * 8;
    const uint8_t *unit = wbase + 16 * threadIdx.x;   // kWinBytes / 16 == kT
    const uint8_t *body_end = c.body + c.body_len;
    // the 16-byte unit is readable when it overlaps the allocation (an aligned
    // unit never crosses a page); bytes at or after the end of the body read as 0
    const uint8_t *alloc_lo = c.in_alloc - ((uintptr_t)c.in_alloc & 15);
    const uint8_t *alloc_hi = c.in_alloc + c.in_alloc_len;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (unit >= alloc_lo && unit < alloc_hi && unit < body_end) {
        v = *reinterpret_cast<const uint4 *>(unit);
        if (unit + 16 > body_end) {
            uint32_t keep = (uint32_t)(body_end - unit);  // 1..15 valid bytes
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int valid = (int)keep - 4 * k;  // bytes of this word that are valid
                if (valid <= 0) w[k] = 0;
                else if (valid < 4) w[k] &= (1u << (8 * valid)) - 1;
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    reinterpret_cast<uint4 *>(L.win)[threadIdx.x] = v;
    if (threadIdx.x < 2) L.win[kWinWords + threadIdx.x] = 0;
}

// Read the 14-bit dynamic header + code lengths (src/decompress/mod.rs:403-507)
// and build the three tables.  Runs on one thread; reads bits through the
// staged window, which the caller guarantees covers the whole header
// (a header is at most 14 + 19*3 + 320*(7+7) bits < 600 bytes).
__device__ __noinline__ int read_dynamic_header(StreamLds &L, uint32_t *bp, int64_t win_bit0, uint64_t end_bit)
{
    uint32_t b = *bp;
    auto avail = [&](uint32_t need) { return (uint64_t)(win_bit0 + b + need) <= end_bit; };
    if (!avail(14)) return RSX_SHORT_INPUT;
    uint32_t bits = fetch_bits(L.win, b);
    uint32_t num_litlen = 257 + (bits & 0x1F);
    uint32_t num_offset = 1 + ((bits >> 5) & 0x1F);
    uint32_t num_precode = 4 + ((bits >> 10) & 0xF);
    b += 14;
    uint32_t i;
    for (i = 0; i < num_precode; i++) {
        if (!avail(3)) return RSX_SHORT_INPUT;
        L.precode_lens[c_precode_perm[i]] = fetch_bits(L.win, b) & 7;
        b += 3;
    }
    for (; i < 19; i++) L.precode_lens[c_precode_perm[i]] = 0;
    if (!build_decode_table(L.precode, L.precode_lens, 19, KIND_PRECODE, kPrecodeBits, 7,
                            L.sorted_syms, nullptr))
        return RSX_BAD_DATA;
    const uint32_t total = num_litlen + num_offset;
    i = 0;
    while (i < total) {
        bits = fetch_bits(L.win, b);
        uint32_t e = L.precode[bits & ((1u << kPrecodeBits) - 1)];
        uint32_t nbits = e & 0xFF;
        if (!avail(nbits)) return RSX_SHORT_INPUT;
        b += nbits;
        bits >>= nbits;
        uint32_t presym = e >> 16;
        if (presym < 16) {
            L.lens[i++] = (uint8_t)presym;
            continue;
        }
        uint32_t rep, val, xb;
        if (presym == 16) {
            if (i == 0) return RSX_BAD_DATA;
            val = L.lens[i - 1];
            xb = 2;
            rep = 3;
        } else if (presym == 17) {
            val = 0;
            xb = 3;
            rep = 3;
        } else {
            val = 0;
            xb = 7;
            rep = 11;
        }
        if (!avail(xb)) return RSX_SHORT_INPUT;
        rep += bits & ((1u << xb) - 1);
        b += xb;
        // a run passing the end is clamped (src/decompress/mod.rs:462-467)
        while (rep-- && i < total) L.lens[i++] = (uint8_t)val;
    }
    *bp = b;
    if (!build_decode_table(L.offset, L.lens + num_litlen, num_offset, KIND_OFFSET, kOffsetBits, 15,
                            L.sorted_syms, nullptr))
        return RSX_BAD_DATA;
    if (!build_decode_table(L.litlen, L.lens, num_litlen, KIND_LITLEN, kLitlenBits, 15,
                            L.sorted_syms, &L.litlen_tablebits))
        return RSX_BAD_DATA;
    return RSX_SUCCESS;
}

// src/decompress/mod.rs:355-401
__device__ __noinline__ void load_static_codes(StreamLds &L)
{
    uint32_t i = 0;
    for (; i < 144; i++) L.lens[i] = 8;
    for (; i < 256; i++) L.lens[i] = 9;
    for (; i < 280; i++) L.lens[i] = 7;
    for (; i < 288; i++) L.lens[i] = 8;
    for (; i < 320; i++) L.lens[i] = 5;
    build_decode_table(L.offset, L.lens + 288, 32, KIND_OFFSET, kOffsetBits, 15, L.sorted_syms,
                       nullptr);
    build_decode_table(L.litlen, L.lens, 288, KIND_LITLEN, kLitlenBits, 15, L.sorted_syms,
                       &L.litlen_tablebits);
}

// Thread-0 part of a round: block headers and symbol decoding from the staged
// window.  Literals are stored into the ring, matches appended to the queue.
// Order of the checks per symbol follows src/decompress/mod.rs:940-1071.
__device__ __noinline__ void decode_round(StreamLds &L, const StreamCtx &c, int64_t win_bit0)
{
    const uint64_t end_bit = c.body_len * 8;
    uint32_t b = (uint32_t)((int64_t)L.bitpos - win_bit0);
    uint64_t out_pos = L.round_start_lo;
    const uint64_t round_start = out_pos;
    uint32_t nm = 0;
    int status = -1;

    while (status < 0 && b < kWinLimitBits && nm < kMaxMatches &&
           out_pos + 258 <= round_start + kRoundOut) {
        if (!L.in_block) {
            // block header; needs the dynamic header fully staged: restart the
            // round at this position unless the window has room for it
            if (b + 8 * 640 > kWinLimitBits && b >= 128) break;
            if ((uint64_t)(win_bit0 + b + 3) > end_bit) {
                status = RSX_SHORT_INPUT;
                break;
            }
            uint32_t bits = fetch_bits(L.win, b);
            L.final_block = bits & 1;
            uint32_t type = (bits >> 1) & 3;
            b += 3;
            if (type == 0) {
                // stored block: byte-align, LEN / NLEN (src/decompress/mod.rs:282-304)
                uint64_t abs = (uint64_t)(win_bit0 + b);
                uint64_t byte = (abs + 7) >> 3;
                if (byte + 4 > c.body_len) {
                    status = RSX_SHORT_INPUT;
                    break;
                }
                const uint8_t *p = c.body + byte;
                uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
                if (len != (~nlen & 0xFFFF)) {
                    status = RSX_BAD_DATA;
                    break;
                }
                L.stored_len = (int32_t)len;
                L.bitpos = (byte + 4) * 8;
                // the copy itself is done by the whole workgroup
                L.num_matches = nm;
                L.round_end_lo = (uint32_t)out_pos;
                L.status = status;
                return;
            } else if (type == 3) {
                status = RSX_BAD_DATA;
                break;
            } else if (type == 2) {
                int r = read_dynamic_header(L, &b, win_bit0, end_bit);
                if (r != RSX_SUCCESS) {
                    status = r;
                    break;
                }
            } else {
                load_static_codes(L);
            }
            L.in_block = 1;
            continue;
        }

        uint32_t bits = fetch_bits(L.win, b);
        uint32_t e = L.litlen[bits & ((1u << L.litlen_tablebits) - 1)];
        uint32_t used = 0;
        if ((e & (kExceptional | kSubtable)) == (kExceptional | kSubtable)) {
            used = e & 0xFF;
            bits >>= used;
            e = L.litlen[(e >> 16) + (bits & ((1u << ((e >> 8) & 0x3F)) - 1))];
        }
        uint32_t total_bits = e & 0xFF;
        if ((uint64_t)(win_bit0 + b + used + total_bits) > end_bit) {
            status = RSX_SHORT_INPUT;
            break;
        }
        b += used + total_bits;
        if (e & kLiteral) {
            if (out_pos >= c.out_cap) {
                status = RSX_INSUFFICIENT_SPACE;
                break;
            }
            L.ring[((uint32_t)out_pos + c.shift) & kRingMask] = (uint8_t)(e >> 16);
            out_pos++;
            continue;
        }
        if (e & kEndOfBlock) {
            L.in_block = 0;
            if (L.final_block) {
                status = RSX_SUCCESS;
                break;
            }
            continue;
        }
        uint32_t clen = (e >> 8) & 0xFF;
        uint32_t length = (e >> 16) + ((bits >> clen) & ((1u << (total_bits - clen)) - 1));

        bits = fetch_bits(L.win, b);
        e = L.offset[bits & ((1u << kOffsetBits) - 1)];
        used = 0;
        if (e & kSubtable) {
            used = e & 0xFF;
            bits >>= used;
            e = L.offset[(e >> 16) + (bits & ((1u << ((e >> 8) & 0x3F)) - 1))];
        }
        total_bits = e & 0xFF;
        if ((uint64_t)(win_bit0 + b + used + total_bits) > end_bit) {
            status = RSX_SHORT_INPUT;
            break;
        }
        b += used + total_bits;
        clen = (e >> 8) & 0xFF;
        uint32_t dist = (e >> 16) + ((bits >> clen) & ((1u << (total_bits - clen)) - 1));
        if (dist > out_pos) {
            status = RSX_BAD_DATA;
            break;
        }
        if (out_pos + length > c.out_cap) {
            status = RSX_INSUFFICIENT_SPACE;
            break;
        }
        L.matches[nm++] = make_uint2((uint32_t)(out_pos - round_start) | (length << 16), dist);
        out_pos += length;
    }
    L.bitpos = (uint64_t)(win_bit0 + b);
    L.num_matches = nm;
    L.round_end_lo = (uint32_t)out_pos;
    L.status = status;
}

// All lanes: resolve the queued matches of this round.
__device__ void copy_round(StreamLds &L, const StreamCtx &c)
{
    const uint32_t nm = L.num_matches;
    const uint32_t round_start = L.round_start_lo;
    const uint32_t round_end = L.round_end_lo;
    // positions >= lds_lo are still in the ring once the round is complete
    const int64_t lds_lo = (int64_t)round_end - (int64_t)kRing;
    const int tid = threadIdx.x;
    for (uint32_t base = 0; base < nm; base += kT) {
        const uint32_t m = base + tid;
        const bool have = m < nm;
        uint32_t dst = 0, len = 0, dist = 1;
        if (have) {
            uint2 q = L.matches[m];
            dst = round_start + (q.x & 0xFFFF);
            len = q.x >> 16;
            dist = q.y;
        }
        // all bytes below need_end must be final before this match may run
        const uint32_t need_end = dst - dist + (len < dist ? len : dist);
        bool done = !have;
        for (;;) {
            unsigned long long open = __ballot(!done);
            if ((tid & 63) == 0)
                L.first_open[tid >> 6] = open ? base + (tid & ~63) + (uint32_t)__builtin_ctzll(open) : ~0u;
            __syncthreads();
            uint32_t f = L.first_open[0];
            for (int w = 1; w < kT / 64; w++) f = f < L.first_open[w] ? f : L.first_open[w];
            if (f == ~0u) break;
            // every byte below the first unfinished match's destination is final
            const uint32_t hw = round_start + (L.matches[f].x & 0xFFFF);
            if (!done && need_end <= hw) {
                for (uint32_t k = 0; k < len; k++) {
                    int64_t sp = (int64_t)dst + k - dist;
                    uint8_t v = sp >= lds_lo ? L.ring[((uint32_t)sp + c.shift) & kRingMask]
                                             : c.out[sp];
                    L.ring[(dst + k + c.shift) & kRingMask] = v;
                }
                done = true;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// All lanes: write output positions [from, to) from the ring to global memory.
// from is on a 16-byte boundary of the global address (or 0); full units use
// 16-byte stores, the ragged first / last unit byte stores.
__device__ void flush_range(StreamLds &L, const StreamCtx &c, uint32_t from, uint32_t to)
{
    if (to <= from) return;
    const uint32_t g_from = from + c.shift, g_to = to + c.shift;  // = global address mod-16 phase
    const uint32_t u0 = g_from >> 4, u1 = (g_to + 15) >> 4;
    for (uint32_t u = u0 + threadIdx.x; u < u1; u += kT) {
        uint32_t lo = u << 4, hi = lo + 16;
        uint8_t *gp = c.out + (int64_t)lo - (int64_t)c.shift;
        if (lo >= g_from && hi <= g_to) {
            *reinterpret_cast<uint4 *>(gp) = *reinterpret_cast<const uint4 *>(&L.ring[lo & kRingMask]);
        } else {
            uint32_t a = lo > g_from ? lo : g_from, z = hi < g_to ? hi : g_to;
            for (uint32_t g = a; g < z; g++) gp[g - lo] = L.ring[g & kRingMask];
        }
    }
}

struct ContainerInfo {
    uint32_t hdr;      // bytes in front of the DEFLATE body
    uint32_t trailer;  // bytes behind it
    int status;        // -1 = ok so far
};

// zlib / gzip header checks, src/decompress/mod.rs:1079-1095 and :1149-1192
__device__ ContainerInfo parse_container(int format, const uint8_t *in, uint64_t in_len)
{
    ContainerInfo ci{0, 0, -1};
    if (format == RSX_FORMAT_ZLIB) {
        if (in_len < 6) {
            ci.status = RSX_SHORT_INPUT;
            return ci;
        }
        uint32_t hdr = (in[0] << 8) | in[1];
        if (hdr % 31 != 0 || ((hdr >> 8) & 0xF) != 8 || ((hdr >> 12) & 0xF) > 7 || ((hdr >> 5) & 1))
            ci.status = RSX_BAD_DATA;
        ci.hdr = 2;
        ci.trailer = 4;
    } else if (format == RSX_FORMAT_GZIP) {
        if (in_len < 18) {
            ci.status = RSX_SHORT_INPUT;
            return ci;
        }
        if (in[0] != 0x1F || in[1] != 0x8B || in[2] != 8 || (in[3] & 0xE0)) {
            ci.status = RSX_BAD_DATA;
            return ci;
        }
        uint32_t flg = in[3];
        uint64_t h = 10;
        if (flg & 0x04) {
            if (h + 2 > in_len) {
                ci.status = RSX_SHORT_INPUT;
                return ci;
            }
            h += 2 + (in[h] | (in[h + 1] << 8));
        }
        if (flg & 0x08) {
            while (h < in_len && in[h] != 0) h++;
            h++;
        }
        if (flg & 0x10) {
            while (h < in_len && in[h] != 0) h++;
            h++;
        }
        if (flg & 0x02) h += 2;
        if (h + 8 > in_len) {
            ci.status = RSX_SHORT_INPUT;
            return ci;
        }
        ci.hdr = (uint32_t)h;
        ci.trailer = 8;
    }
    return ci;
}

__global__ __launch_bounds__(kT) void inflate_kernel(
    int format, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    uint64_t total_in, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status, uint64_t *__restrict__ body_end)
{
    __shared__ StreamLds L;
    const uint32_t item = blockIdx.x;
    if (item >= n) return;
    const int tid = threadIdx.x;

    const uint64_t ib = in_off[item], ie = in_off[item + 1];
    const uint64_t ob = out_off[item], oe = out_off[item + 1];
    StreamCtx c;
    c.in_alloc = in;
    c.in_alloc_len = total_in;
    c.out = out + ob;
    c.out_cap = oe - ob;
    c.shift = (uint32_t)((uintptr_t)c.out & 15);

    // container framing is parsed redundantly by every lane (a few cached loads)
    ContainerInfo ci = parse_container(format, in + ib, ie - ib);
    if (ci.status >= 0) {
        if (tid == 0) {
            status[item] = ci.status;
            out_sizes[item] = 0;
            body_end[item] = 0;
        }
        return;
    }
    c.body = in + ib + ci.hdr;
    c.body_len = (ie - ib) - ci.hdr - ci.trailer;

    if (tid == 0) {
        L.bitpos = 0;
        L.status = -1;
        L.in_block = 0;
        L.final_block = 0;
        L.stored_len = -1;
        L.round_start_lo = 0;
        L.round_end_lo = 0;
        L.num_matches = 0;
    }
    uint32_t flushed = 0;  // output positions below this are in global memory (uniform)
    __syncthreads();

    for (;;) {
        int64_t win_bit0;
        stage_window(L, c, &win_bit0);
        __syncthreads();
        if (tid == 0) decode_round(L, c, win_bit0);
        __syncthreads();
        copy_round(L, c);
        const uint32_t round_end = L.round_end_lo;
        int st = L.status;
        int32_t stored = L.stored_len;
        __syncthreads();
        if (st < 0 && stored >= 0) {
            // stored block body (src/decompress/mod.rs:305-346): first push out what
            // the round produced, then move the bytes in ring-sized pieces
            uint32_t upto = ((round_end + c.shift) & ~15u);
            upto = upto > c.shift ? upto - c.shift : 0;
            if (upto < flushed) upto = flushed;
            flush_range(L, c, flushed, upto);
            flushed = upto;
            uint64_t src_byte = L.bitpos >> 3;
            uint64_t avail_in = c.body_len - src_byte;
            uint64_t avail_out = c.out_cap - round_end;
            uint32_t len = (uint32_t)stored;
            uint32_t pos = round_end;
            if (len > avail_in || len > avail_out) {
                // whichever side runs dry first; a tie reports the output side
                st = avail_out <= avail_in ? RSX_INSUFFICIENT_SPACE : RSX_SHORT_INPUT;
            } else {
                const uint8_t *sp = c.body + src_byte;
                for (uint32_t done = 0; done < len;) {
                    uint32_t piece = len - done < kRoundOut ? len - done : kRoundOut;
                    for (uint32_t k = tid; k < piece; k += kT)
                        L.ring[(pos + k + c.shift) & kRingMask] = sp[done + k];
                    __syncthreads();
                    pos += piece;
                    done += piece;
                    uint32_t up2 = ((pos + c.shift) & ~15u);
                    up2 = up2 > c.shift ? up2 - c.shift : 0;
                    if (up2 < flushed) up2 = flushed;
                    flush_range(L, c, flushed, up2);
                    flushed = up2;
                    __syncthreads();
                }
                if (L.final_block) st = RSX_SUCCESS;
            }
            __syncthreads();
            if (tid == 0) {
                L.stored_len = -1;
                L.bitpos += (uint64_t)len * 8;
                L.round_start_lo = pos;
                L.round_end_lo = pos;
                L.status = st;
            }
            __syncthreads();
            if (st >= 0) break;
            continue;
        }
        if (st >= 0) break;
        uint32_t upto = ((round_end + c.shift) & ~15u);
        upto = upto > c.shift ? upto - c.shift : 0;
        if (upto < flushed) upto = flushed;
        flush_range(L, c, flushed, upto);
        flushed = upto;
        if (tid == 0) L.round_start_lo = round_end;
        __syncthreads();
    }

    const int st = L.status;
    const uint32_t produced = L.round_end_lo;
    if (st == RSX_SUCCESS) flush_range(L, c, flushed, produced);
    if (tid == 0) {
        status[item] = st;
        out_sizes[item] = st == RSX_SUCCESS ? produced : 0;
        // first byte behind the DEFLATE stream, relative to the item start
        body_end[item] = ci.hdr + ((L.bitpos + 7) >> 3);
    }
}

// zlib / gzip trailer check (src/decompress/mod.rs:1104-1120, :1202-1233): the
// checksum of the produced bytes must match the stored one (and ISIZE for gzip).
__global__ void verify_trailer_kernel(int format, const uint8_t *__restrict__ in,
                                      const uint64_t *__restrict__ in_off, uint32_t n,
                                      const uint64_t *__restrict__ body_end,
                                      const uint32_t *__restrict__ sums,
                                      uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || status[i] != RSX_SUCCESS) return;
    const uint8_t *t = in + in_off[i] + body_end[i];
    bool ok;
    if (format == RSX_FORMAT_ZLIB) {
        uint32_t want = ((uint32_t)t[0] << 24) | (t[1] << 16) | (t[2] << 8) | t[3];
        ok = want == sums[i];
    } else {
        uint32_t want = t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24);
        uint32_t isize = t[4] | (t[5] << 8) | (t[6] << 16) | ((uint32_t)t[7] << 24);
        ok = want == sums[i] && isize == (uint32_t)out_sizes[i];
    }
    if (!ok) {
        status[i] = RSX_BAD_DATA;
        out_sizes[i] = 0;
    }
}

}  // namespace

size_t inflate_ws_bytes(size_t n, uint64_t total_out)
{
    return align_up(n * sizeof(uint64_t), 256) + align_up(n * sizeof(uint32_t), 256) +
           checksum_ws_bytes(n, total_out) + 1024;
}

int inflate_launch(rsx_engine *e, int format, const uint8_t *in, const uint64_t *in_off, size_t n,
                   uint64_t total_in, uint64_t total_out, uint64_t max_out, uint8_t *out,
                   const uint64_t *out_off, uint64_t *out_sizes, int32_t *status)
{
    if (n == 0) return RSX_OK;
    if (n > 0x7FFFFFFFull || max_out >= (1ull << 32)) {
        e->err = "batch too large (items must be < 4 GiB, n < 2^31)";
        return RSX_E_INVALID;
    }
    uint64_t *body_end = static_cast<uint64_t *>(ws_take(e, n * sizeof(uint64_t)));
    uint32_t *sums = static_cast<uint32_t *>(ws_take(e, n * sizeof(uint32_t)));
    if (!body_end || !sums) {
        e->err = "workspace exhausted";
        return RSX_E_NOMEM;
    }
    (void)hipEventRecord(e->ev_begin, e->stream);
    inflate_kernel<<<(uint32_t)n, kT, 0, e->stream>>>(format, in, in_off, (uint32_t)n, total_in, out,
                                                      out_off, out_sizes, status, body_end);
    (void)hipEventRecord(e->ev_end, e->stream);
    e->timed = true;
    RSX_HIP(e, hipGetLastError());
    if (format != RSX_FORMAT_DEFLATE) {
        int kind = format == RSX_FORMAT_GZIP ? RSX_CHECKSUM_CRC32 : RSX_CHECKSUM_ADLER32;
        uint32_t init = format == RSX_FORMAT_GZIP ? 0u : 1u;
        // failed items have out_sizes 0 and cost nothing
        int r = checksum_launch(e, kind, init, out, out_off, out_sizes, n, total_out, max_out, sums);
        if (r != RSX_OK) return r;
        verify_trailer_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, e->stream>>>(
            format, in, in_off, (uint32_t)n, body_end, sums, out_sizes, status);
        RSX_HIP(e, hipGetLastError());
    }
    return RSX_OK;
}
