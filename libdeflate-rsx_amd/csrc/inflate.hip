// inflate.hip -- batch raw-DEFLATE / zlib / gzip decompression on gfx950.
//
// GPU counterpart of BatchDecompressor::decompress_batch (src/batch.rs:74-101)
// and of the per-item decoder it drives: block state machine
// (src/decompress/mod.rs:213-353), dynamic header (:403-507), decode-table
// build (:1324-1495), Huffman block body + LZ copy (:509-1072), zlib / gzip
// framing (:1074-1240).  Result codes are the reference's DecompressResult.
//
// One 256-thread workgroup decodes one stream in rounds:
//   1. stage   a ~4 KiB window of compressed bytes into LDS with 16-byte
//              coalesced loads;
//   2. decode  all 256 lanes decode Huffman symbols at once.  The window is
//              cut into 128-bit sub-sequences, lane i starts at a guessed bit
//              position inside sub-sequence i and decodes until it crosses
//              into sub-sequence i+1.  Huffman streams re-synchronise, so after
//              a few passes of "restart where the previous lane really ended"
//              every lane's start is the true one (lane 0 always is).  A prefix
//              sum of the bytes each lane produces gives its output position,
//              then each lane decodes once more and emits: literals straight
//              into a 32 KiB LDS ring that mirrors the tail of the output,
//              matches into a queue as (dst, len, dist);
//   3. copy    all lanes resolve the queued matches in parallel -- a match may
//              run once every byte below its source end is final; sources older
//              than the ring are read back from the already written output;
//   4. flush   the round's bytes leave the ring as 16-byte coalesced stores.
// Block headers and decode tables are still handled by one lane.
// HBM traffic is therefore one read of the compressed bytes and one write of
// the output; the history window is served from LDS (and L2 for far matches).
//
// Roofline: HBM-bound by design, algorithmic bytes per stream =
// compressed_len + uncompressed_len.
#include "engine.h"

#include <cstdlib>
#include <vector>

namespace {

constexpr int kT = 256;                       // threads per stream
constexpr int kRingBits = 15;
constexpr uint32_t kRing = 1u << kRingBits;   // LDS mirror of the newest output bytes
constexpr uint32_t kRingMask = kRing - 1;
constexpr uint32_t kWinBytes = 4096 + 256;    // staged compressed bytes per round
constexpr uint32_t kWinWords = kWinBytes / 4;
// a token may start only while two dwords past its last fetch are still staged:
// litlen fetch at b, offset fetch at most 20 bits later, each reads words w, w+1
constexpr uint32_t kWinLimitBits = (kWinWords - 2) * 32 - 20;
constexpr uint32_t kSubBits = 128;            // compressed bits per lane and round
// one lane decodes at most kSubBits/2 matches of 258 bytes (a match token is
// at least 2 bits), so a round that commits at least lane 0 always fits
constexpr uint32_t kRoundOut = 20480;         // output bytes per round (<= kRing)
constexpr uint32_t kMaxMatches = 2048;        // match queue entries per round
static_assert(kSubBits / 2 * 258 <= kRoundOut && kSubBits / 2 <= kMaxMatches, "lane 0 must fit");
static_assert(kRoundOut <= kRing, "a round may not lap the ring");

// lane exit word: bit position (window relative) | flags
constexpr uint32_t kExitEob = 1u << 31;       // the block ended in this lane
constexpr uint32_t kExitStop = 1u << 30;      // ran out of input inside a token
constexpr uint32_t kExitPosMask = (1u << 20) - 1;

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
    uint32_t match_dl[kMaxMatches];  // (dst - round_start) | len << 15
    uint16_t match_dist[kMaxMatches];  // dist - 1
    uint32_t lane_exit[kT];
    uint32_t scan_carry[2 * (kT / 64)];
    uint32_t emit_error;          // min over lanes of lane << 8 | DecompressResult
    uint32_t stat_spec, stat_copy; // pass counters (STATS builds)
    uint32_t first_flag;          // first lane whose sub-sequence ends the block / the input
    uint16_t sorted_syms[288];
    uint8_t lens[288 + 32 + 140];
    uint8_t precode_lens[20];
    uint32_t first_open[kT / 64];
    uint32_t done_bits[kT / 32];  // finished matches of the current copy batch
    // control block written by thread 0, read by all after a barrier
    uint32_t litlen_tablebits;
    uint32_t num_matches;
    uint32_t round_start_lo, round_end_lo;  // output positions (streams are < 4 GiB)
    int32_t status;       // -1 while running, else a DecompressResult
    int32_t in_block;     // 1 while inside a Huffman block body
    int32_t final_block;
    int32_t stored_len;   // >= 0: a stored block of that many bytes is pending
    int32_t restage;      // header parse wants the window moved first
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
    const uint8_t *body_end = c.body + c.body_len;
    // a 16-byte unit is readable when it overlaps the allocation (an aligned
    // unit never crosses a page); bytes at or after the end of the body read as 0
    const uint8_t *alloc_lo = c.in_alloc - ((uintptr_t)c.in_alloc & 15);
    const uint8_t *alloc_hi = c.in_alloc + c.in_alloc_len;
    for (uint32_t u = threadIdx.x; u < kWinBytes / 16; u += kT) {
        const uint8_t *unit = wbase + 16 * (size_t)u;
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
        reinterpret_cast<uint4 *>(L.win)[u] = v;
    }
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

// Thread-0 part of a round: when the stream is between blocks, read the next
// block header from the staged window (src/decompress/mod.rs:225-252) and build
// the decode tables.  Leaves L.in_block / L.stored_len / L.status / L.restage set.
__device__ __noinline__ void header_step(StreamLds &L, const StreamCtx &c, int64_t win_bit0)
{
    const uint64_t end_bit = c.body_len * 8;
    uint32_t b = (uint32_t)((int64_t)L.bitpos - win_bit0);
    L.restage = 0;
    // a dynamic header is at most 14 + 19*3 + 320*14 bits; have it fully staged
    if (b + 8 * 640 > kWinLimitBits && b >= 128) {
        L.restage = 1;
        return;
    }
    if ((uint64_t)(win_bit0 + b + 3) > end_bit) {
        L.status = RSX_SHORT_INPUT;
        return;
    }
    uint32_t bits = fetch_bits(L.win, b);
    L.final_block = bits & 1;
    uint32_t type = (bits >> 1) & 3;
    b += 3;
    if (type == 0) {
        // stored block: byte-align, LEN / NLEN (src/decompress/mod.rs:282-304)
        uint64_t byte = ((uint64_t)(win_bit0 + b) + 7) >> 3;
        if (byte + 4 > c.body_len) {
            L.status = RSX_SHORT_INPUT;
            return;
        }
        const uint8_t *p = c.body + byte;
        uint32_t len = p[0] | (p[1] << 8), nlen = p[2] | (p[3] << 8);
        if (len != (~nlen & 0xFFFF)) {
            L.status = RSX_BAD_DATA;
            return;
        }
        L.stored_len = (int32_t)len;
        L.bitpos = (byte + 4) * 8;
        return;
    }
    if (type == 3) {
        L.status = RSX_BAD_DATA;
        return;
    }
    if (type == 2) {
        int r = read_dynamic_header(L, &b, win_bit0, end_bit);
        if (r != RSX_SUCCESS) {
            L.status = r;
            return;
        }
    } else {
        load_static_codes(L);
    }
    L.in_block = 1;
    L.bitpos = (uint64_t)(win_bit0 + b);
}

// One decoded token.  kind: 0 literal, 1 match, 2 end of block.
struct Token {
    uint32_t nbits;  // bits consumed by the whole token
    uint32_t kind;
    uint32_t value;  // literal byte, or match length
    uint32_t dist;
};

// Per-lane bit buffer over the staged window: at least 32 valid bits after
// refill(), one LDS dword read per 32 consumed bits.
struct BitReader {
    uint64_t bb;     // unread bits, LSB first
    uint32_t bc;     // valid bits in bb
    uint32_t wnext;  // next window dword to load

    __device__ inline void init(const StreamLds &L, uint32_t b)
    {
        const uint32_t w = b >> 5;
        bb = (((uint64_t)L.win[w + 1] << 32) | L.win[w]) >> (b & 31);
        bc = 64 - (b & 31);
        wnext = w + 2;
    }
    __device__ inline void refill(const StreamLds &L)
    {
        if (bc < 32) {
            bb |= (uint64_t)L.win[wnext++] << bc;
            bc += 32;
        }
    }
    __device__ inline void consume(uint32_t n)
    {
        bb >>= n;
        bc -= n;
    }
};

// Decode the token at the reader's position.  Bits past the end of the body
// read as zero; the caller checks nbits against the bits really available.
__device__ inline Token decode_token(const StreamLds &L, BitReader &r, uint32_t litlen_mask)
{
    Token t;
    r.refill(L);
    uint32_t bits = (uint32_t)r.bb;
    uint32_t e = L.litlen[bits & litlen_mask];
    uint32_t used = 0;
    if ((e & (kExceptional | kSubtable)) == (kExceptional | kSubtable)) {
        used = e & 0xFF;
        bits >>= used;
        e = L.litlen[(e >> 16) + (bits & ((1u << ((e >> 8) & 0x3F)) - 1))];
    }
    uint32_t total_bits = e & 0xFF;
    used += total_bits;
    r.consume(used);
    t.nbits = used;
    t.dist = 0;
    if (e & kLiteral) {
        t.kind = 0;
        t.value = (e >> 16) & 0xFF;
        return t;
    }
    if (e & kEndOfBlock) {
        t.kind = 2;
        t.value = 0;
        return t;
    }
    uint32_t clen = (e >> 8) & 0xFF;
    t.kind = 1;
    t.value = (e >> 16) + ((bits >> clen) & ((1u << (total_bits - clen)) - 1));
    r.refill(L);
    bits = (uint32_t)r.bb;
    e = L.offset[bits & ((1u << kOffsetBits) - 1)];
    used = 0;
    if (e & kSubtable) {
        used = e & 0xFF;
        bits >>= used;
        e = L.offset[(e >> 16) + (bits & ((1u << ((e >> 8) & 0x3F)) - 1))];
    }
    total_bits = e & 0xFF;
    clen = (e >> 8) & 0xFF;
    t.dist = (e >> 16) + ((bits >> clen) & ((1u << (total_bits - clen)) - 1));
    used += total_bits;
    r.consume(used);
    t.nbits += used;
    return t;
}

// What one lane learns from decoding its sub-sequence without emitting.
struct Scan {
    uint32_t exit;   // where the next lane starts (window bit) | kExit* flags
    uint32_t bytes;  // output bytes of the tokens that start in [start, bound)
    uint32_t nm;     // matches among them
};

// Decode tokens that start in [start, bound); end_b = first bit past the body.
__device__ inline Scan scan_span(const StreamLds &L, uint32_t start, uint32_t bound, int64_t end_b,
                                 uint32_t litlen_mask)
{
    Scan r{start, 0, 0};
    uint32_t b = start;
    BitReader br;
    br.init(L, b);
    while (b < bound) {
        Token t = decode_token(L, br, litlen_mask);
        if ((int64_t)(b + t.nbits) > end_b) {
            r.exit = b | kExitStop;
            return r;
        }
        b += t.nbits;
        if (t.kind == 0) {
            r.bytes += 1;
        } else if (t.kind == 1) {
            r.bytes += t.value;
            r.nm += 1;
        } else {
            r.exit = b | kExitEob;
            return r;
        }
    }
    r.exit = b;
    return r;
}

// exclusive prefix sums of two 32-bit values over the workgroup
__device__ inline void wg_exclusive_scan2(StreamLds &L, uint32_t a, uint32_t b, uint32_t *ea,
                                          uint32_t *eb)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t xa = a, xb = b;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t ya = __shfl_up(xa, o), yb = __shfl_up(xb, o);
        if (lane >= o) {
            xa += ya;
            xb += yb;
        }
    }
    if (lane == 63) {
        L.scan_carry[2 * w] = xa;
        L.scan_carry[2 * w + 1] = xb;
    }
    __syncthreads();
    uint32_t ca = 0, cb = 0;
    for (int k = 0; k < w; k++) {
        ca += L.scan_carry[2 * k];
        cb += L.scan_carry[2 * k + 1];
    }
    *ea = ca + xa - a;
    *eb = cb + xb - b;
    __syncthreads();
}

// All lanes: decode the Huffman block body in the staged window, starting at
// L.bitpos.  On return literals are in the ring, matches in the queue and the
// control block (bitpos, round_end, num_matches, in_block, status) is updated.
// Order of the per-token checks follows src/decompress/mod.rs:940-1071.
__device__ void decode_round_parallel(StreamLds &L, const StreamCtx &c, int64_t win_bit0)
{
    const int tid = threadIdx.x;
    const uint32_t b0 = (uint32_t)((int64_t)L.bitpos - win_bit0);
    const int64_t end_b = (int64_t)(c.body_len * 8) - win_bit0;  // first bit past the body
    const uint32_t litlen_mask = (1u << L.litlen_tablebits) - 1;
    const uint32_t round_start = L.round_start_lo;
    // lane i owns tokens that start in [b0 + i*S, b0 + (i+1)*S), clipped to the window
    const uint32_t lo = b0 + tid * kSubBits;
    uint32_t bound = lo + kSubBits;
    if (bound > kWinLimitBits) bound = kWinLimitBits;

    // ---- passes 1..k: find every lane's true start ------------------------------
    // A lane that decodes an end-of-block (or runs out of input) while its start is
    // still a guess must not silence the lanes behind it: they keep their own guess
    // until the chain in front of them is settled.
    uint32_t start = lo;          // lane 0's guess is exact
    Scan sc{0, 0, 0};
    bool need = true;
    // the true prefix grows by at least one lane per pass, so kT passes always suffice
    for (int pass = 0; pass <= kT; pass++) {
        if (need) {
            if (start >= bound) {
                sc = Scan{start, 0, 0};  // nothing starts here (window exhausted or overshoot)
            } else {
                sc = scan_span(L, start, bound, end_b, litlen_mask);
            }
            L.lane_exit[tid] = sc.exit;
        }
        __syncthreads();
        uint32_t ns = lo;
        if (tid > 0) {
            uint32_t pe = L.lane_exit[tid - 1];
            if (!(pe & (kExitEob | kExitStop))) ns = pe;
        }
        need = ns != start;
        start = ns;
        if (tid == 0) L.stat_spec++;
        if (!__syncthreads_or(need)) break;
    }
    // every start now follows from lane 0's: the first lane that ended the block
    // (or the input) is real, and nothing behind it belongs to this block
    if (tid == 0) {
        L.first_flag = kT;
        L.emit_error = ~0u;
    }
    __syncthreads();
    if (sc.exit & (kExitEob | kExitStop)) atomicMin(&L.first_flag, (uint32_t)tid);
    __syncthreads();
    const bool live = (uint32_t)tid <= L.first_flag;
    if (!live) {
        sc.bytes = 0;
        sc.nm = 0;
    }

    // ---- output positions and the committed prefix ---------------------------------
    uint32_t ex_bytes, ex_nm;
    wg_exclusive_scan2(L, sc.bytes, sc.nm, &ex_bytes, &ex_nm);
    const bool commit = live && ex_bytes + sc.bytes <= kRoundOut && ex_nm + sc.nm <= kMaxMatches;
    // `commit` is monotone over lanes; the last committed lane publishes the round
    L.lane_exit[tid] = commit ? sc.exit | 0x10000000u : sc.exit;
    __syncthreads();
    const bool last = commit && (tid == kT - 1 || !(L.lane_exit[tid + 1] & 0x10000000u));

    // ---- emit -------------------------------------------------------------------------
    if (commit) {
        uint32_t b = start;
        uint64_t op = (uint64_t)round_start + ex_bytes;
        uint32_t mi = ex_nm;
        int err = -1;
        BitReader br;
        br.init(L, b);
        while (b < bound) {
            Token t = decode_token(L, br, litlen_mask);
            if ((int64_t)(b + t.nbits) > end_b) {
                err = RSX_SHORT_INPUT;
                break;
            }
            b += t.nbits;
            if (t.kind == 0) {
                if (op >= c.out_cap) {
                    err = RSX_INSUFFICIENT_SPACE;
                    break;
                }
                L.ring[((uint32_t)op + c.shift) & kRingMask] = (uint8_t)t.value;
                op++;
            } else if (t.kind == 1) {
                if (t.dist > op) {
                    err = RSX_BAD_DATA;
                    break;
                }
                if (op + t.value > c.out_cap) {
                    err = RSX_INSUFFICIENT_SPACE;
                    break;
                }
                L.match_dl[mi] = (uint32_t)(op - round_start) | (t.value << 15);
                L.match_dist[mi] = (uint16_t)(t.dist - 1);
                mi++;
                op += t.value;
            } else {
                break;
            }
        }
        if (err >= 0) atomicMin(&L.emit_error, ((uint32_t)tid << 8) | (uint32_t)err);
    }
    __syncthreads();
    if (last) {
        const uint32_t ee = L.emit_error;
        L.bitpos = (uint64_t)(win_bit0 + (int64_t)(sc.exit & kExitPosMask));
        L.round_end_lo = round_start + ex_bytes + sc.bytes;
        L.num_matches = ex_nm + sc.nm;
        if (ee != ~0u) {
            L.status = (int32_t)(ee & 0xFF);  // the first failing token in stream order
        } else if (sc.exit & kExitEob) {
            L.in_block = 0;
            if (L.final_block) L.status = RSX_SUCCESS;
        }
    }
    __syncthreads();
}

// All lanes: resolve the queued matches of this round, kT at a time.  Inside a
// batch a match waits only for the earlier matches whose destination overlaps its
// source bytes (found by binary search in the queue, which is sorted by
// destination); everything older is final.  Passes per batch = depth of the
// dependency chain, not the number of near matches.
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
        uint32_t dep_a = 0, dep_b = 0;  // batch-relative range of matches my source overlaps
        if (tid < kT / 32) L.done_bits[tid] = 0;
        if (have) {
            uint32_t q = L.match_dl[m];
            const uint32_t drel = q & 0x7FFF;
            dst = round_start + drel;
            len = q >> 15;
            dist = (uint32_t)L.match_dist[m] + 1;
            // source bytes [s, s_end) relative to round_start (s may lie before the round)
            const int32_t s = (int32_t)drel - (int32_t)dist;
            const int32_t s_end = s + (int32_t)(len < dist ? len : dist);
            if (s_end > 0) {
                // first match in [base, m) whose end is behind s
                uint32_t lo = base, hi = m;
                while (lo < hi) {
                    uint32_t mid = (lo + hi) >> 1;
                    uint32_t qq = L.match_dl[mid];
                    if ((int32_t)((qq & 0x7FFF) + (qq >> 15)) > s) hi = mid; else lo = mid + 1;
                }
                dep_a = lo - base;
                uint32_t j = lo;
                while (j < m && (int32_t)(L.match_dl[j] & 0x7FFF) < s_end) j++;
                dep_b = j - base;
            }
        }
        bool done = !have, published = !have;
        __syncthreads();
        // the first unfinished match of a batch can always run, so kT passes suffice
        for (int pass = 0; pass <= kT; pass++) {
            if (!done) {
                bool ready = true;
                for (uint32_t j = dep_a; j < dep_b; j++)
                    if (!((L.done_bits[j >> 5] >> (j & 31)) & 1)) {
                        ready = false;
                        break;
                    }
                if (ready) {
                    const uint32_t d0 = dst + c.shift, s0 = d0 - dist;
                    if ((int64_t)dst - dist < lds_lo) {
                        // far source: part of it has left the ring, read that from the output
                        for (uint32_t k = 0; k < len; k++) {
                            int64_t sp = (int64_t)dst + k - dist;
                            uint8_t v = sp >= lds_lo ? L.ring[((uint32_t)sp + c.shift) & kRingMask]
                                                     : c.out[sp];
                            L.ring[(d0 + k) & kRingMask] = v;
                        }
                    } else if (dist >= 8) {
                        // 8 independent loads, then 8 stores: one LDS round trip per 8 bytes
                        uint32_t k = 0;
                        for (; k + 8 <= len; k += 8) {
                            uint8_t v[8];
#pragma unroll
                            for (int j = 0; j < 8; j++) v[j] = L.ring[(s0 + k + j) & kRingMask];
#pragma unroll
                            for (int j = 0; j < 8; j++) L.ring[(d0 + k + j) & kRingMask] = v[j];
                        }
                        uint8_t v[8];
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (k + j < len) v[j] = L.ring[(s0 + k + j) & kRingMask];
#pragma unroll
                        for (int j = 0; j < 8; j++)
                            if (k + j < len) L.ring[(d0 + k + j) & kRingMask] = v[j];
                    } else {
                        // overlapping copy: the output repeats the last `dist` bytes
                        uint64_t pat = 0;
#pragma unroll
                        for (uint32_t j = 0; j < 7; j++)
                            if (j < dist) pat |= (uint64_t)L.ring[(s0 + j) & kRingMask] << (8 * j);
                        uint32_t idx = 0;
                        for (uint32_t k = 0; k < len; k++) {
                            L.ring[(d0 + k) & kRingMask] = (uint8_t)(pat >> (8 * idx));
                            idx = idx + 1 == dist ? 0 : idx + 1;
                        }
                    }
                    done = true;
                    dep_a = dep_b = 0;
                }
            }
            if (tid == 0) L.stat_copy++;
            __syncthreads();  // copies of this pass are visible before their done bits
            if (done && !published) {
                atomicOr(&L.done_bits[tid >> 5], 1u << (tid & 31));
                published = true;
            }
            if (!__syncthreads_or(!done)) break;
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

// STATS builds add per-stream phase timers (development aid, RSX_INFLATE_STATS=1):
// stats[item*8 + k]: 0 rounds, 1 speculation passes, 2 copy passes, 3..7 cycles in
// stage / header / decode / copy / flush
template <bool STATS>
__global__ __launch_bounds__(kT) void inflate_kernel(
    int format, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    uint64_t total_in, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status, uint64_t *__restrict__ body_end,
    unsigned long long *__restrict__ stats)
{
    unsigned long long tS = 0, tH = 0, tD = 0, tC = 0, tF = 0, t0 = 0, nRounds = 0;
#define STAT_BEGIN() do { if (STATS) t0 = clock64(); } while (0)
#define STAT_END(acc) do { if (STATS) acc += clock64() - t0; } while (0)
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
        L.restage = 0;
        L.emit_error = ~0u;
        L.stat_spec = 0;
        L.stat_copy = 0;
        L.round_start_lo = 0;
        L.round_end_lo = 0;
        L.num_matches = 0;
    }
    uint32_t flushed = 0;  // output positions below this are in global memory (uniform)
    uint64_t seen_bitpos = ~0ull;  // progress watchdog (uniform)
    int stalled = 0;
    __syncthreads();

    for (;;) {
        // every round either consumes input or moves the window; anything else is a bug
        // and must not spin on the GPU
        {
            const uint64_t bp = L.bitpos;
            stalled = bp == seen_bitpos ? stalled + 1 : 0;
            seen_bitpos = bp;
            __syncthreads();
            if (stalled > 4) {
                if (tid == 0) L.status = RSX_BAD_DATA;
                __syncthreads();
                break;
            }
        }
        int64_t win_bit0;
        STAT_BEGIN();
        stage_window(L, c, &win_bit0);
        __syncthreads();
        STAT_END(tS);
        nRounds++;
        if (!L.in_block) {
            STAT_BEGIN();
            __syncthreads();
            if (tid == 0) header_step(L, c, win_bit0);
            __syncthreads();
            STAT_END(tH);
        }
        {
            // snapshot the control block, then let nobody change it before all have read it
            const int restage = L.restage, st0 = L.status, stored0 = L.stored_len;
            const uint32_t b0 = (uint32_t)((int64_t)L.bitpos - win_bit0);
            __syncthreads();
            if (restage) {  // the window has to move to the block header first
                if (tid == 0) L.restage = 0;
                continue;
            }
            if (st0 < 0 && stored0 < 0) {
                // keep a whole sub-sequence plus one token inside the window
                if (b0 + kSubBits + 64 > kWinLimitBits) continue;
                STAT_BEGIN();
                decode_round_parallel(L, c, win_bit0);
                STAT_END(tD);
                // a failed round leaves holes in the match queue: nothing to copy
                STAT_BEGIN();
                if (L.status < 0 || L.status == RSX_SUCCESS) copy_round(L, c);
                STAT_END(tC);
            }
        }
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
        STAT_BEGIN();
        uint32_t upto = ((round_end + c.shift) & ~15u);
        upto = upto > c.shift ? upto - c.shift : 0;
        if (upto < flushed) upto = flushed;
        flush_range(L, c, flushed, upto);
        flushed = upto;
        if (tid == 0) L.round_start_lo = round_end;
        __syncthreads();
        STAT_END(tF);
    }

    const int st = L.status;
    const uint32_t produced = L.round_end_lo;
    if (st == RSX_SUCCESS) flush_range(L, c, flushed, produced);
    if (tid == 0) {
        status[item] = st;
        out_sizes[item] = st == RSX_SUCCESS ? produced : 0;
        // first byte behind the DEFLATE stream, relative to the item start
        body_end[item] = ci.hdr + ((L.bitpos + 7) >> 3);
        if (STATS) {
            unsigned long long *q = stats + (size_t)item * 8;
            q[0] = nRounds; q[1] = L.stat_spec; q[2] = L.stat_copy;
            q[3] = tS; q[4] = tH; q[5] = tD; q[6] = tC; q[7] = tF;
        }
    }
#undef STAT_BEGIN
#undef STAT_END
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
    const bool want_stats = getenv("RSX_INFLATE_STATS") != nullptr;
    unsigned long long *d_stats = nullptr;
    if (want_stats) {
        RSX_HIP(e, hipMalloc(&d_stats, n * 8 * sizeof(unsigned long long)));
    }
    (void)hipEventRecord(e->ev_begin, e->stream);
    if (want_stats)
        inflate_kernel<true><<<(uint32_t)n, kT, 0, e->stream>>>(
            format, in, in_off, (uint32_t)n, total_in, out, out_off, out_sizes, status, body_end, d_stats);
    else
        inflate_kernel<false><<<(uint32_t)n, kT, 0, e->stream>>>(
            format, in, in_off, (uint32_t)n, total_in, out, out_off, out_sizes, status, body_end, nullptr);
    (void)hipEventRecord(e->ev_end, e->stream);
    if (want_stats) {
        std::vector<unsigned long long> h(n * 8);
        RSX_HIP(e, hipMemcpy(h.data(), d_stats, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_stats);
        unsigned long long sum[8] = {0};
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 8; k++) sum[k] += h[i * 8 + k];
        fprintf(stderr,
                "[rsx inflate stats] streams %zu rounds %.1f spec_passes/round %.2f copy_passes/round %.2f "
                "cycles/stream: stage %.0f header %.0f decode %.0f copy %.0f flush %.0f\n",
                n, (double)sum[0] / n, (double)sum[1] / (double)sum[0], (double)sum[2] / (double)sum[0],
                (double)sum[3] / n, (double)sum[4] / n, (double)sum[5] / n, (double)sum[6] / n,
                (double)sum[7] / n);
    }
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
