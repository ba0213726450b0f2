// inflate.hip -- batch raw-DEFLATE / zlib / gzip decompression on gfx950.
//
// GPU counterpart of BatchDecompressor::decompress_batch (src/batch.rs:74-101)
// and of the per-item decoder it drives: block state machine
// (src/decompress/mod.rs:213-353), dynamic header (:403-507), decode-table
// build (:1324-1495), Huffman block body + LZ copy (:509-1072), zlib / gzip
// framing (:1074-1240).  Result codes are the reference's DecompressResult.
//
// A batch is decoded by two kernels, one 256-thread workgroup per stream (or per part of a stream) in each:
//   inflate_decode_kernel  Huffman decoding only.  Rounds over a 15 KiB LDS window of the compressed bytes;
//              the window is cut into 256 sub-sequences of up to 480 bits, every lane starts at a guessed
//              token boundary (the end of a 1024-bit warm-up decode in front of its sub-sequence) and decodes
//              until it crosses into the next one; passes of "restart where the lane in front really ended"
//              converge because Huffman streams re-synchronise.  Tokens (a literal, or length and distance)
//              go to token rows in device memory, [row k][lane]: the k-th token of every sub-sequence.  Block
//              headers are read and the decode tables built by the first wave, all 64 lanes taking part.
//   inflate_expand_kernel  LZ77 expansion only.  Groups of sub-sequences that fit a 16 KiB LDS ring are walked
//              from their checkpoints in parallel (literals into the ring, matches into a queue), then resolved
//              2048 output bytes at a time by pointer jumping over per-byte source pointers (overlapping and
//              chained matches resolve in log steps), and leave the ring as 16-byte coalesced stores.
// Few, large streams are first cut at the sync-flush joints the reference's chunked compressor writes
// (part_* kernels) so that every 256 KiB chunk is a workgroup's worth of work of its own.
// inflate_kernel is the older one-kernel decoder (stage / decode / copy / flush rounds in one workgroup, 31 KB of
// LDS): it takes whatever the two kernels give up -- damaged streams, slots that are too small, streams that run
// out of token rows -- decodes it again from the start and reports the reference's status code, and it serves
// rsx_batch_decompress_device_noalloc, whose static scratch holds no token rows.
// zlib / gzip: a CRC-32 / Adler-32 pass over the produced bytes (checksum.hip) and verify_trailer_kernel.
//
// Roofline: HBM by design (algorithmic bytes per stream = compressed_len + uncompressed_len); the token rows
// cost 4 bytes per token written and read, and both kernels are bound by vector-ALU issue, not by memory.
#include "engine.h"

#include <cstdlib>
#include <type_traits>
#include <vector>

// build with -DRSX_DEV_TIMERS=1 to get the finer in-kernel timers of RSX_INFLATE_STATS
#ifndef RSX_DEV_TIMERS
#define RSX_DEV_TIMERS 0
#endif
#ifndef RSX_DBG
#define RSX_DBG 0   // development checks
#endif
#ifndef RSX_TABLE_CHECK
#define RSX_TABLE_CHECK 0   // 1: the table build fails when it has left an entry unwritten (development)
#endif
#ifndef RSX_TWO_BARRIERS
#define RSX_TWO_BARRIERS 0   // 1: the copy phase's rounds separate their reads from their writes (for comparison)
#endif

namespace {

// -DRSX_SPLIT_TIMERS=1: phase timers of the two kernels (development aid), summed over all streams:
// 0 stage 1 header 2 warm-up 3 first scan 4 later passes 5 round tail 6 rounds 7 passes
// 8 expand: records+scan 9 walk 10 copy 11 flush 12 groups 13, 14 header read / tables
// 15 copy: pointer scatter 16 rounds 17 gather 18 number of rounds 19 number of steps
#ifndef RSX_SPLIT_TIMERS
#define RSX_SPLIT_TIMERS 0
#endif
// A slot is only ever added to by one thread of a workgroup; the sums are kept in LDS (L.tacc) and go
// to global memory once, when the workgroup ends (an atomic per sample would be timed itself).
#if RSX_SPLIT_TIMERS
static __device__ unsigned long long g_split_t[24];
#define SPLIT_T_ADD(k, v) (L.tacc[k] += (unsigned long long)(v))
#define SPLIT_T_NOW() clock64()
#define SPLIT_T_FIELD unsigned long long tacc[24];
#define SPLIT_T_BEGIN() do { if (threadIdx.x < 24) L.tacc[threadIdx.x] = 0; } while (0)
#define SPLIT_T_END() do { WG_SYNC(); if (threadIdx.x < 24) atomicAdd(&g_split_t[threadIdx.x], L.tacc[threadIdx.x]); } while (0)
#else
#define SPLIT_T_ADD(k, v) do { } while (0)
#define SPLIT_T_NOW() 0ull
#define SPLIT_T_FIELD
#define SPLIT_T_BEGIN() do { } while (0)
#define SPLIT_T_END() do { } while (0)
#endif


constexpr int kT = 256;                       // threads per stream
constexpr int kRingBits = 14;
constexpr uint32_t kRing = 1u << kRingBits;   // LDS mirror of the newest output bytes
constexpr uint32_t kRingMask = kRing - 1;
constexpr uint32_t kWinBytes = 4096;          // staged compressed bytes per round of the one-kernel decoder: one 16-byte load per lane
constexpr uint32_t kWinWords = kWinBytes / 4;
// a token may start only while two dwords past its last fetch are still staged:
// litlen fetch at b, offset fetch at most 20 bits later, each reads words w, w+1
constexpr uint32_t win_limit_bits(uint32_t win_words) { return (win_words - 2) * 32 - 20; }
constexpr uint32_t kWinLimitBits = win_limit_bits(kWinWords);
#ifndef RSX_SUB_BITS
#define RSX_SUB_BITS 126
#endif
constexpr uint32_t kSubBits = RSX_SUB_BITS;   // compressed bits per lane and round
// one lane decodes at most kSubBits/2 matches of 258 bytes (a match token is
// at least 2 bits), so a round that commits at least lane 0 always fits
constexpr uint32_t kRoundOut = 16384 - 16;    // output bytes per round
constexpr uint32_t kMaxMatches = 1792;        // match queue entries per round
#ifndef RSX_WARM_BITS
#define RSX_WARM_BITS 720
#endif
constexpr uint32_t kWarmBits = RSX_WARM_BITS;  // warm-up in front of a sub-sequence's first guess
constexpr uint32_t kCopySub = 2048;           // output bytes resolved per copy step (u16 pointers fit in win)
static_assert(kCopySub * 2 <= kWinBytes, "copy pointers live in the window buffer");
#ifndef RSX_EXPERIMENT
static_assert(kSubBits / 2 * 258 <= kRoundOut && kSubBits / 2 <= kMaxMatches, "lane 0 must fit");
#endif
// up to 15 bytes of a round stay unflushed (16-byte store granularity): the next round
// must not overwrite them in the ring
static_assert(kRoundOut + 16 <= kRing, "a round may not lap the unflushed tail in the ring");

// lane exit word: bit position (window relative) | flags
constexpr uint32_t kExitEob = 1u << 31;       // the block ended in this lane
constexpr uint32_t kExitStop = 1u << 30;      // ran out of input inside a token
constexpr uint32_t kExitPosMask = (1u << 20) - 1;

constexpr int kLitlenBits = 10;
constexpr int kLitlenEnough = 1334;           // enough(288, 10, 15): largest table any complete code needs
constexpr int kOffsetBits = 8;
constexpr int kOffsetEnough = 402;            // enough(32, 8, 15)
static_assert(kOffsetEnough <= 512, "the offset table sits in front of the litlen table at index 512");
constexpr int kPrecodeBits = 7;
constexpr int kPrecodeEnough = 128;

// Decode-table entries of the litlen and offset tables.  They hold what the entries of
// src/decompress/tables.rs:32-53 hold (symbol value, codeword length, codeword + extra bits)
// and, in addition, the table the NEXT lookup uses, so that a decoder step derives its next
// state from the entry alone:
//   bits  0-4   bits the step consumes (codeword + extra bits; for a pointer: the root index)
//   bits  5-8   codeword bits among them (the rest are extra bits)
//   bits  9-10  what the step finishes: kTokNone / kTokEob / kTokLit / kTokMatch
//   bit  11     the next lookup is the litlen root (the flag is that table's byte offset, 4 * kTabLitlen)
//   bits 12-15  index width of the next lookup
//   bits 16-30  literal / length base / distance base / byte offset of the subtable
//   bit  31     subtable pointer
constexpr uint32_t kTokNone = 0, kTokEob = 1, kTokLit = 2, kTokMatch = 3;
constexpr uint32_t kEntNextLit = 1u << 11;
constexpr uint32_t kEntPointer = 1u << 31;
constexpr uint32_t kTabLitlen = 512;          // the offset table sits at table index 0, the litlen table here
static_assert(4 * kTabLitlen == kEntNextLit, "the flag is the table's byte offset");

enum { KIND_LITLEN = 1, KIND_OFFSET = 2 };

// Base values and extra-bit counts of the length symbols 257.. and of the offset symbols (RFC 1951 3.2.5;
// the symbols that no valid stream uses decode like the reference's tables say).  They are computed, not
// loaded: a table in constant memory costs the table build one memory latency per codeword length.
constexpr uint16_t kLenBase[31] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,
                                   17, 19, 23, 27, 31, 35, 43, 51, 59,  67,  83,
                                   99, 115, 131, 163, 195, 227, 258, 258, 258};
constexpr uint8_t kLenExtra[31] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2,
                                   3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0};
constexpr uint16_t kOffBase[32] = {1,    2,    3,    4,    5,    7,     9,     13,
                                   17,   25,   33,   49,   65,   97,    129,   193,
                                   257,  385,  513,  769,  1025, 1537,  2049,  3073,
                                   4097, 6145, 8193, 12289, 16385, 24577, 24577, 24577};
constexpr uint8_t kOffExtra[32] = {0, 0, 0, 0, 1, 1, 2,  2,  3,  3,  4,  4,  5,  5,  6,  6,
                                   7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 13, 13};
__host__ __device__ constexpr uint32_t len_extra(uint32_t i) { return i < 8 || i >= 28 ? 0u : (i - 4) >> 2; }
__host__ __device__ constexpr uint32_t len_base(uint32_t i)
{
    return i >= 28 ? 258u : i < 8 ? 3 + i : 3 + ((4 + (i & 3)) << len_extra(i));
}
__host__ __device__ constexpr uint32_t off_extra(uint32_t s) { return s < 4 ? 0u : ((s < 29 ? s : 29u) - 2) >> 1; }
__host__ __device__ constexpr uint32_t off_base(uint32_t s)
{
    return s < 4 ? 1 + s : 1 + ((2 + ((s < 29 ? s : 29u) & 1)) << off_extra(s));
}
constexpr bool symbol_formulas_match_tables()
{
    for (uint32_t i = 0; i < 31; i++)
        if (len_base(i) != kLenBase[i] || len_extra(i) != kLenExtra[i]) return false;
    for (uint32_t s = 0; s < 32; s++)
        if (off_base(s) != kOffBase[s] || off_extra(s) != kOffExtra[s]) return false;
    return true;
}
static_assert(symbol_formulas_match_tables(), "the formulas give the tables' values");

// per-symbol decode results, src/decompress/tables.rs:10-30, :55-379; `len` = codeword bits
// this entry accounts for, `root_bits` = index width of the litlen root table
__device__ inline uint32_t make_entry(int kind, uint32_t sym, uint32_t len, uint32_t root_bits)
{
    const uint32_t back = kEntNextLit | (root_bits << 12);   // the next token starts at the litlen root
    if (kind == KIND_LITLEN) {
        if (sym < 256) return (sym << 16) | (kTokLit << 9) | back | (len << 5) | len;
        if (sym == 256) return (kTokEob << 9) | back | (len << 5) | len;
        // a length: its distance comes from the offset root
        return (len_base(sym - 257) << 16) | (kTokNone << 9) | ((uint32_t)kOffsetBits << 12) | (len << 5) |
               (len + len_extra(sym - 257));
    }
    return (off_base(sym) << 16) | (kTokMatch << 9) | back | (len << 5) | (len + off_extra(sym));
}

// the precode's entries: symbol << 16 | (codeword length + extra bits of the symbol) << 8 | codeword length
__device__ inline uint32_t make_precode_entry(uint32_t sym, uint32_t len)
{
    const uint32_t xb = sym < 16 ? 0u : sym == 16 ? 2u : sym == 17 ? 3u : 7u;
    return (sym << 16) | ((len + xb) << 8) | len;
}

// Everything a workgroup keeps in LDS.  Three groups of arrays are never live at
// the same time and share storage: header parsing scratch, the per-sub-sequence
// bookkeeping of the speculation passes, and the match queue.
struct HeaderScratch {
    uint32_t precode[kPrecodeEnough];
    uint16_t sorted_syms[288];
    uint8_t lens[288 + 32 + 140];
    uint32_t num_litlen, num_offset;   // alphabet sizes of the block whose tables are being built
    uint32_t plan;                     // 0 nothing to build, 1 the static code, 3 a dynamic header whose counts start at `after`
    uint32_t after;                    // window bit position behind the header
};
struct SpecScratch {
    uint32_t sub_start[kT];       // start handed to a sub-sequence that is re-decoded
    uint16_t sub_bytes[kT], sub_nm[kT];
    uint16_t work[kT];            // compacted list of sub-sequences to re-decode
    uint32_t lane_exit[kT];       // where each sub-sequence ended (dead once the emit pass fills the queue)
};
struct TokSpecScratch {           // the speculation passes of the token decoder
    uint32_t sub_start[kT];       // start handed to a sub-sequence that is re-decoded
    uint32_t info[kT];            // tokens | matches << 8 | bytes << 16 of each sub-sequence
    uint16_t work[kT];            // compacted list of sub-sequences to re-decode
    uint32_t lane_exit[kT];       // where each sub-sequence ended
};
struct MatchQueue {
    // Only where each match starts.  Its distance and length wait in the ring, in the first three
    // of the bytes the match will produce (dist - 1 low, high, len - 3): nobody reads those bytes
    // as data before the copy phase has written them.
    uint16_t dst[kMaxMatches];    // dst - round_start
};
template <uint32_t kWinBytesT, uint32_t kRingT>
struct __align__(16) StreamLdsT {
    static constexpr uint32_t kWinBytes = kWinBytesT, kWinWords = kWinBytesT / 4;
    static constexpr uint32_t kWinLimitBits = win_limit_bits(kWinWords);
    uint32_t win[kWinWords + 4];      // staged input (first: its LDS address is 0, so indexing it costs no add);
                                      // reused as u16 source pointers by the copy phase
    uint8_t ring[kRingT];
    uint32_t tab[kTabLitlen + kLitlenEnough];   // decode tables: offset at index 0, litlen at kTabLitlen
    union {
        HeaderScratch hdr;
        SpecScratch spec;
        TokSpecScratch tspec;
        MatchQueue mq;
    };
    uint32_t scan_carry[3 * (kT / 64) + 4];   // scans and flags; [8..11]: the copy phase's spanning match, then the flags of odd rounds
    uint32_t emit_error;          // min over lanes of lane << 8 | DecompressResult
    uint32_t stat_spec, stat_copy; // pass counters (STATS builds)
    uint32_t first_flag;          // first lane whose sub-sequence ends the block / the input
    unsigned long long stat_t[4]; // development timers: first pass, later passes, prefix, emit
    // control block written by one lane, read by all after a barrier
    uint32_t litlen_tablebits;
    uint32_t num_matches;
    uint32_t round_start_lo, round_end_lo;  // output positions (streams are < 4 GiB)
    int32_t status;       // -1 while running, else a DecompressResult
    int32_t in_block;     // 1 while inside a Huffman block body
    int32_t final_block;
    int32_t stored_len;   // >= 0: a stored block of that many bytes is pending
    int32_t restage;      // header parse wants the window moved first
    uint32_t pool_chunk;  // (token decoder) 0, or 1 + the row of the shared pool where the stream's current chunk of rows starts
    uint64_t bitpos;      // next unread bit, relative to the start of the DEFLATE body
    SPLIT_T_FIELD
};

// The serial header code sits behind non-inlined calls.  Its LDS operands carry their
// address space in the type: through plain pointers every access would be a flat one.
typedef StreamLdsT<kWinBytes, kRing> StreamLds;   // the one-kernel decoder's
#define RSX_LDS __attribute__((address_space(3)))
typedef RSX_LDS StreamLds LdsStream;

// Orders the LDS traffic of one wave: its lanes run in step and the LDS serves a wave's
// requests in order, so this only has to keep the compiler from moving accesses across.
#define WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// The litlen or offset table (src/decompress/mod.rs:1324-1495: same checks, same root and
// subtable layout, the entries in this file's format) built by the 64 lanes of one wave.  The wave is
// one of many on its CU and gets an instruction issued every twenty-odd cycles, so the build is shaped to
// need few wave instructions, not little work per lane:
// * symbols stay in symbol order, 64 to a batch.  Four ballots over the bits of the lengths give every
//   lane the set of lanes with its length (its rank among them is the canonical order) and lanes 0..15 the
//   count of "their" length; the counts of the sixteen lengths, the first codeword of each and the index
//   of its first symbol live in lanes 0..15 of three registers, read by lane number;
// * a codeword no longer than the root index writes its own root entries (the canonical codeword,
//   bit-reversed, is the low end of the index; every value of the bits above it is a replica); the few
//   codewords with more than eight replicas are written by all lanes together;
// * longer codewords are put in canonical order (LDS) and taken 64 at a time: which lengths occur under a
//   codeword's root prefix follows from the first codewords and counts alone, so every lane knows its
//   subtable's size without looking at its neighbours, and a prefix sum over "first codeword of its
//   subtable" lanes gives the subtables' places -- the order and the sizes the serial build arrives at.
// `table_index` is where table[0] sits in the decoder's table space (subtable pointers are absolute byte
// offsets); `root_bits` the index width of the litlen root, for the entries that lead back to it (ignored
// when table_bits_ret is given: then this IS the litlen table and its own shrunk width is used and returned).
__device__ __noinline__ bool build_decode_table_wave(RSX_LDS uint32_t *table, uint32_t table_index,
                                                     const RSX_LDS uint8_t *lens, uint32_t num_syms, int kind,
                                                     uint32_t table_bits, uint32_t root_bits,
                                                     RSX_LDS uint16_t *sorted_syms, RSX_LDS uint32_t *table_bits_ret)
{
    constexpr uint32_t kBatches = 5;   // 288 symbols at most
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1;
    num_syms = __builtin_amdgcn_readfirstlane(num_syms);
    table_bits = __builtin_amdgcn_readfirstlane(table_bits);
    root_bits = __builtin_amdgcn_readfirstlane(root_bits);
#if RSX_TABLE_CHECK
    for (uint32_t i = lane; i < (kind == KIND_LITLEN ? (uint32_t)kLitlenEnough : (uint32_t)kOffsetEnough); i += 64) table[i] = 0;
    WAVE_SYNC();
#endif
    uint32_t len_k[kBatches], rank_k[kBatches];   // of symbol 64 k + lane: its length, its rank among the symbols of that length
    uint32_t cntv = 0;      // lane l < 16: symbols of length l
    uint32_t one_sym = 0;   // a symbol of length 1
#pragma unroll
    for (uint32_t k = 0; k < kBatches; k++) {
        len_k[k] = 0;
        rank_k[k] = 0;
        if (k * 64 < num_syms) {   // (uniform)
            const uint32_t s = k * 64 + lane;
            const bool valid = s < num_syms;
            const uint32_t mylen = valid ? (uint32_t)lens[s] & 15u : 0u;
            const unsigned long long all = __ballot(valid);
            unsigned long long own = all, idx = all;   // lanes with this lane's length / with length (lane & 15)
#pragma unroll
            for (uint32_t bit = 0; bit < 4; bit++) {
                const unsigned long long has = __ballot((mylen >> bit & 1) != 0);
                own &= (mylen >> bit & 1) ? has : ~has;
                idx &= (lane >> bit & 1) ? has : ~has;
            }
            rank_k[k] = (uint32_t)__shfl((int)cntv, (int)mylen) + (uint32_t)__popcll(own & below);
            len_k[k] = mylen;
            cntv += (uint32_t)__popcll(idx);
            const unsigned long long ones = __ballot(mylen == 1);
            if (ones) one_sym = k * 64 + (uint32_t)__ffsll((long long)ones) - 1;
        }
    }
    auto cnt = [&](uint32_t len) { return (uint32_t)__builtin_amdgcn_readlane((int)cntv, (int)len); };
    uint32_t max_codeword_len = 1;
    {
        const unsigned long long used = __ballot(lane >= 1 && lane < 16 && cntv != 0);
        if (used) max_codeword_len = 63u - (uint32_t)__clzll((long long)used);
    }
    if (table_bits_ret) {
        if (table_bits > max_codeword_len) table_bits = max_codeword_len;
        if (lane == 0) *table_bits_ret = table_bits;
        root_bits = table_bits;
    }
    // lane l of fcv: the canonical (MSB-first) first codeword of length l; of startv: how many codewords are shorter
    uint32_t fcv = 0, startv = 0, codespace_used = 0, num_coded = 0;
    {
        uint32_t fc = 0;
        for (uint32_t len = 1; len <= max_codeword_len; len++) {
            fcv = lane == len ? fc : fcv;
            startv = lane == len ? num_coded : startv;
            const uint32_t c = cnt(len);
            num_coded += c;
            codespace_used = (codespace_used << 1) + c;
            fc = (fc + c) << 1;
        }
    }
    if (codespace_used > (1u << max_codeword_len)) return false;
    if (codespace_used < (1u << max_codeword_len)) {
        uint32_t s;
        if (codespace_used == 0) {
            s = 0;
        } else {
            if (codespace_used != (1u << (max_codeword_len - 1)) || cnt(1) != 1) return false;
            s = one_sym;
        }
        const uint32_t e = make_entry(kind, s, 1, root_bits);
        for (uint32_t i = lane; i < (1u << table_bits); i += 64) table[i] = e;
        WAVE_SYNC();
        return true;
    }

    // codewords that fit the root index
    const uint32_t root_coded = max_codeword_len > table_bits
                                    ? (uint32_t)__builtin_amdgcn_readlane((int)startv, (int)(table_bits + 1)) : num_coded;
#pragma unroll
    for (uint32_t k = 0; k < kBatches; k++) {
        if (k * 64 >= num_syms) continue;   // (uniform)
        const uint32_t len = len_k[k], sym = k * 64 + lane;
        const uint32_t code = (uint32_t)__shfl((int)fcv, (int)len) + rank_k[k];
        const uint32_t rev = len ? __brev(code) >> (32 - len) : 0u;
        const bool in_root = len != 0 && len <= table_bits;
        const uint32_t e = make_entry(kind, sym, in_root ? len : 1u, root_bits);
        if (in_root && len + 3 >= table_bits) {
            for (uint32_t rep = 0; rep < (1u << (table_bits - len)); rep++) table[rev | rep << len] = e;
        }
        // the few short codewords have many replicas: all lanes write those of one
        unsigned long long wide = __ballot(in_root && len + 3 < table_bits);
        while (wide) {   // (uniform)
            const uint32_t j = (uint32_t)__ffsll((long long)wide) - 1;
            wide &= wide - 1;
            const uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)j);
            const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)j);
            const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rev, (int)j);
            for (uint32_t rep = lane; rep < (1u << (table_bits - lj)); rep += 64) table[rj | rep << lj] = ej;
        }
        // longer codewords wait in canonical order (the lane exchange stands outside the branch: a lane that is
        // switched off hands nothing on)
        const uint32_t first_of_len = (uint32_t)__shfl((int)startv, (int)len);
        if (len > table_bits) sorted_syms[first_of_len - root_coded + rank_k[k]] = (uint16_t)(sym | len << 9);
    }
    WAVE_SYNC();
    if (max_codeword_len <= table_bits) {   // no codeword is longer than the root index
#if RSX_TABLE_CHECK
        bool hole = false;
        for (uint32_t i = lane; i < (1u << table_bits); i += 64) hole = hole || (table[i] & 31) == 0;
        if (__any(hole)) return false;
#endif
        return true;
    }

    const uint32_t num_long = num_coded - root_coded;
    uint32_t slots_before = 0;   // entries of the subtables in front of this batch's
    for (uint32_t j0 = 0; j0 < num_long; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool valid = j < num_long;
        const uint32_t packed = valid ? (uint32_t)sorted_syms[j] : (uint32_t)((table_bits + 1) << 9);
        const uint32_t sym = packed & 511, len = packed >> 9;
        const uint32_t code = (uint32_t)__shfl((int)fcv, (int)len) + (j + root_coded - (uint32_t)__shfl((int)startv, (int)len));
        const uint32_t prefix = code >> (len - table_bits);   // the codeword's first table_bits bits
        // the shortest and the longest codeword under this prefix, and the first of the shortest ones
        uint32_t minl = 0, maxl = 0, first_code = 0;
        for (uint32_t l = table_bits + 1; l <= max_codeword_len; l++) {   // (uniform)
            const uint32_t fc = (uint32_t)__builtin_amdgcn_readlane((int)fcv, (int)l), c = cnt(l);
            if (c == 0) continue;
            const uint32_t lo = max(fc, prefix << (l - table_bits)), hi = min(fc + c, (prefix + 1) << (l - table_bits));
            if (lo < hi) {
                first_code = minl ? first_code : lo;
                minl = minl ? minl : l;
                maxl = l;
            }
        }
        const uint32_t sub_bits = maxl - table_bits;   // (a valid lane's own codeword lies under its prefix: maxl >= len)
        const bool head = valid && len == minl && code == first_code;
        uint32_t incl = head ? 1u << sub_bits : 0u;
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += y;
        }
        incl += slots_before;
        slots_before = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (valid) {
            const uint32_t start = (1u << table_bits) + incl - (1u << sub_bits);
            const uint32_t rev = __brev(code) >> (32 - len);
            const uint32_t e = make_entry(kind, sym, len - table_bits, root_bits);
            const uint32_t stride = 1u << (len - table_bits);
            for (uint32_t i = start + (rev >> table_bits); i < start + (1u << sub_bits); i += stride) table[i] = e;
            if (head)
                table[rev & ((1u << table_bits) - 1)] = kEntPointer | (4 * (table_index + start) << 16) |
                                                        (sub_bits << 12) | (table_bits << 5) | table_bits;
        }
    }
    WAVE_SYNC();
#if RSX_TABLE_CHECK
    {   // development: an entry that nothing wrote would stall the decoder
        bool hole = false;
        for (uint32_t i = lane; i < (1u << table_bits) + slots_before; i += 64) hole = hole || (table[i] & 31) == 0;
        if (__any(hole)) return false;
    }
#endif
    return true;
}

// 32 valid bits starting at window bit offset b
template <class WordPtr>
__device__ inline uint32_t fetch_bits(WordPtr win, uint32_t b)
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
template <class LS, int kThreads = kT>
__device__ inline void stage_window(LS &L, const StreamCtx &c, int64_t *win_bit0)
{
    constexpr uint32_t kWinBytes = LS::kWinBytes, kWinWords = LS::kWinWords;
    const uint64_t byte_pos = L.bitpos >> 3;
    const uint8_t *cur = c.body + byte_pos;
    const uint8_t *wbase = cur - ((uintptr_t)cur & 15);
    *win_bit0 = ((int64_t)(wbase - c.body)) * 8;
    const uint8_t *body_end = c.body + c.body_len;
    // a 16-byte unit is readable when it overlaps the allocation (an aligned
    // unit never crosses a page); bytes at or after the end of the body read as 0
    const uint8_t *alloc_lo = c.in_alloc - ((uintptr_t)c.in_alloc & 15);
    const uint8_t *alloc_hi = c.in_alloc + c.in_alloc_len;
    constexpr uint32_t kUnits = kWinBytes / 16, kPerLane = (kUnits + kThreads - 1) / kThreads;
    uint4 v[kPerLane];
#pragma unroll
    for (uint32_t i = 0; i < kPerLane; i++) {   // (every load is issued before the first one is needed)
        const uint32_t u = threadIdx.x + i * kThreads;
        const uint8_t *unit = wbase + 16 * (size_t)u;
        v[i] = make_uint4(0, 0, 0, 0);
        if (u < kUnits && unit >= alloc_lo && unit < alloc_hi && unit < body_end)
            v[i] = *reinterpret_cast<const uint4 *>(unit);
    }
#pragma unroll
    for (uint32_t i = 0; i < kPerLane; i++) {
        const uint32_t u = threadIdx.x + i * kThreads;
        const uint8_t *unit = wbase + 16 * (size_t)u;
        if (unit < body_end && unit + 16 > body_end) {
            uint32_t keep = (uint32_t)(body_end - unit);  // 1..15 valid bytes
            uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                int valid = (int)keep - 4 * k;  // bytes of this word that are valid
                if (valid <= 0) w[k] = 0;
                else if (valid < 4) w[k] &= (1u << (8 * valid)) - 1;
            }
            v[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (u < kUnits) reinterpret_cast<uint4 *>(L.win)[u] = v[i];
    }
    if (threadIdx.x < 4) L.win[kWinWords + threadIdx.x] = 0;
}

// The 14-bit dynamic header + code lengths (src/decompress/mod.rs:403-507) go into L.hdr.lens, the two
// alphabet sizes into L.hdr.num_litlen / num_offset; the big tables are built afterwards by the whole
// wave.  Bits are read through the staged window, which the caller guarantees covers the whole header
// (a header is at most 14 + 19*3 + 320*(7+7) bits < 600 bytes) and 40 bytes behind it.
struct HeaderResult {
    int code;      // DecompressResult
    uint32_t b;    // window bit position behind the header (by value: no pointer into the caller's stack)
};
// A dynamic header (src/decompress/mod.rs:403-478) is read by the first wave, every lane taking part and
// every value that decides control flow uniform.  First the three counts and the code lengths of the
// precode, then the precode's decode table (src/decompress/mod.rs:1324-1495 for 19 symbols and codewords of
// at most 7 bits = the table index, so no subtables; same checks, same entries): lane s holds the length of
// symbol s and works out its canonical codeword from the ballots of the seven lengths, lanes t and t + 64
// pick the symbol whose bit-reversed codeword ends table indices t and t + 64.
template <class LS>
__device__ __noinline__ HeaderResult read_dynamic_preamble_wave(LS &L, uint32_t b, int64_t win_bit0, uint64_t end_bit)
{
    const uint32_t lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1;
    b = __builtin_amdgcn_readfirstlane(b);
    const uint64_t left = end_bit - (uint64_t)win_bit0;   // (the window never starts behind the end)
    const uint32_t end_b = __builtin_amdgcn_readfirstlane(left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left);
    if (b + 14 > end_b) return HeaderResult{RSX_SHORT_INPUT, b};
    const uint32_t bits = __builtin_amdgcn_readfirstlane(fetch_bits(L.win, b));
    const uint32_t num_litlen = 257 + (bits & 0x1F);
    const uint32_t num_offset = 1 + ((bits >> 5) & 0x1F);
    const uint32_t num_precode = 4 + ((bits >> 10) & 0xF);
    b += 14;
    if (b + 3 * num_precode > end_b) return HeaderResult{RSX_SHORT_INPUT, b};
    // field i is the length of symbol perm[i] (RFC 1951 3.2.7); lane s takes field inv[s], 5 bits each
    // inv = {3, 17, 15, 13, 11, 9, 7, 5, 4, 6, 8, 10, 12, 14, 16, 18, 0, 1, 2}
    constexpr unsigned long long kInvLo = 3ull | 17ull << 5 | 15ull << 10 | 13ull << 15 | 11ull << 20 | 9ull << 25 |
                                          7ull << 30 | 5ull << 35 | 4ull << 40 | 6ull << 45 | 8ull << 50 | 10ull << 55;
    constexpr unsigned long long kInvHi = 12ull | 14ull << 5 | 16ull << 10 | 18ull << 15 | 0ull << 20 | 1ull << 25 | 2ull << 30;
    const uint32_t field = lane < 12 ? (uint32_t)(kInvLo >> (5 * lane)) & 31 : lane < 19 ? (uint32_t)(kInvHi >> (5 * (lane - 12))) & 31 : 31u;
    uint32_t mylen = 0;
    if (field < num_precode) mylen = fetch_bits(L.win, b + 3 * field) & 7;
    b += 3 * num_precode;

    uint32_t cnt[8], max_len = 1, rank = 0;
#pragma unroll
    for (uint32_t len = 1; len <= 7; len++) {
        const unsigned long long m = __ballot(mylen == len);
        cnt[len] = (uint32_t)__popcll(m);
        if (mylen == len) rank = (uint32_t)__popcll(m & below);
        if (cnt[len]) max_len = len;
    }
    uint32_t codespace_used = 0, first_code[8], code = 0;
#pragma unroll
    for (uint32_t len = 1; len <= 7; len++) {
        if (len <= max_len) codespace_used = (codespace_used << 1) + cnt[len];
        first_code[len] = code;
        code = (code + cnt[len]) << 1;
    }
    if (codespace_used > (1u << max_len)) return HeaderResult{RSX_BAD_DATA, b};
    uint32_t e0, e1;
    if (codespace_used < (1u << max_len)) {
        uint32_t sym = 0;
        if (codespace_used != 0) {
            if (codespace_used != (1u << (max_len - 1)) || cnt[1] != 1) return HeaderResult{RSX_BAD_DATA, b};
            sym = (uint32_t)__ffsll((long long)__ballot(mylen == 1)) - 1;   // the only symbol with a codeword
        }
        e0 = e1 = make_precode_entry(sym, 1);
    } else {
        uint32_t mycode = 0;
#pragma unroll
        for (uint32_t len = 1; len <= 7; len++)
            if (mylen == len) mycode = first_code[len] + rank;
        const uint32_t rev = mylen ? __brev(mycode) >> (32 - mylen) : 0u;
        const uint32_t word = mylen | rev << 8;
        e0 = e1 = make_precode_entry(0, 1);   // (a complete code leaves none of them: every index ends one codeword)
        for (uint32_t sym = 0; sym < 19; sym++) {
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)sym);
            const uint32_t len = w & 0xFF, mask = (1u << len) - 1;
            if (len == 0) continue;   // (uniform)
            if ((lane & mask) == w >> 8) e0 = make_precode_entry(sym, len);
            if (((lane + 64) & mask) == w >> 8) e1 = make_precode_entry(sym, len);
        }
    }
    L.hdr.precode[lane] = e0;
    L.hdr.precode[lane + 64] = e1;
    if (lane == 0) {
        L.hdr.num_litlen = num_litlen;
        L.hdr.num_offset = num_offset;
    }
    WAVE_SYNC();
    return HeaderResult{RSX_SUCCESS, b};
}

// The code lengths of a dynamic header (src/decompress/mod.rs:436-478), read by one wave.  A wave that walks
// the tokens of the precode one after the other is bound by how fast one wave can issue instructions (a
// few hundred cycles per token when the other waves of the CU are busy), so the tokens are decoded side by
// side instead: lane k takes the tokens that start in bits [32k, 32k + 32) behind `b`.  Where its first
// token starts is where lane k - 1 stopped; lanes guess, decode, hand their exits on and decode again
// until nothing changes (a Huffman decoder that starts at a wrong bit falls into step with the right one
// after a few tokens, so two or three passes do).  Then a prefix sum over the lanes' run totals tells
// every lane where its first length goes, and a last pass writes them.  Only non-zero lengths are
// written (the array is cleared first), so no run is longer than six.
// The reference's checks come out the same: its ShortInput cases are all "a token ends behind the
// input", and since the tokens follow each other that is the case exactly when the last token does; its one
// BadData case (a repeat with nothing in front of it) can only be the very first token.
template <class LS>
__device__ __noinline__ HeaderResult read_code_lengths_wave(LS &L, uint32_t b, int64_t win_bit0, uint64_t end_bit)
{
    constexpr uint32_t kSeg = 32;      // bits per lane
    constexpr uint32_t kNone = 0xFF;   // "no length yet" in the scan of last values
    constexpr uint32_t kLastWord = LS::kWinWords + 1;   // (the window has four zero words behind its data)
    const uint32_t lane = threadIdx.x & 63;
    // (told to the compiler as uniform)
    const uint32_t total = __builtin_amdgcn_readfirstlane(L.hdr.num_litlen + L.hdr.num_offset);
    b = __builtin_amdgcn_readfirstlane(b);
    // window bit position of the end of the input, as far as 32 bits hold it
    const uint64_t left = end_bit - (uint64_t)win_bit0;   // (the window never starts behind the end)
    const uint32_t end_b = __builtin_amdgcn_readfirstlane(left > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)left);
    {
        RSX_LDS uint32_t *lw = (RSX_LDS uint32_t *)L.hdr.lens;   // 320 lengths
        lw[lane] = 0;
        if (lane < 16) lw[64 + lane] = 0;
    }
    // one token: its bits at `pos`, counted from the lane's segment start, out of the lane's 64-bit piece
    struct Tok { uint32_t sym, adv, rep; };
    auto token = [&](uint64_t piece, uint32_t off) {
        const uint32_t bits = (uint32_t)(piece >> off);
        const uint32_t e = L.hdr.precode[bits & ((1u << kPrecodeBits) - 1)];
        const uint32_t nb = e & 0xFF, adv = (e >> 8) & 0xFF, sym = e >> 16;
        const uint32_t xv = (bits >> nb) & ((1u << (adv - nb)) - 1);
        return Tok{sym, adv, sym < 16 ? 1u : (sym == 18 ? 11u : 3u) + xv};
    };
    // (only how far the token at `off` reaches)
    auto token_adv = [&](uint64_t piece, uint32_t off) {
        return (L.hdr.precode[(uint32_t)(piece >> off) & ((1u << kPrecodeBits) - 1)] >> 8) & 0xFF;
    };
    uint32_t i_base = 0, carry_last = 0;
    for (bool first_window = true;; first_window = false) {
        // the lane's 64 bits from its segment start on: every token that starts in the segment, or up to 13
        // bits behind it, lies inside with its 14 bits
        const uint32_t seg0 = b + lane * kSeg;
        uint64_t piece;
        {
            const uint32_t w = seg0 >> 5;
            const uint32_t w0 = L.win[w < kLastWord ? w : kLastWord], w1 = L.win[w + 1 < kLastWord ? w + 1 : kLastWord],
                           w2 = L.win[w + 2 < kLastWord ? w + 2 : kLastWord];
            const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, seg0 & 31), hi = __builtin_amdgcn_alignbit(w2, w1, seg0 & 31);
            piece = lo | (uint64_t)hi << 32;
        }
        uint32_t start = 0, exit_off = 0;   // offsets from seg0
        bool need = true;
        for (uint32_t pass = 0; pass < 64; pass++) {
            if (need) {
                uint32_t off = start;
                while (off < kSeg) off += token_adv(piece, off);
                exit_off = off;
            }
            // lane k + 1 starts where lane k stopped
            const uint32_t handed = (uint32_t)__shfl_up((int)exit_off, 1) - kSeg;
            need = lane > 0 && handed != start;
            if (need) start = handed;
            if (!__any(need)) break;
        }
        // the lane's tokens: how many lengths they stand for, and the last length they set
        uint32_t reps = 0, lastv = kNone;
        for (uint32_t off = start; off < kSeg;) {
            const Tok t = token(piece, off);
            reps += t.rep;
            lastv = t.sym < 16 ? t.sym : t.sym == 16 ? lastv : 0u;
            off += t.adv;
        }
        // where the lane's first length goes, and the length in front of it
        uint32_t incl = reps, lastscan = lastv;
        for (uint32_t o = 1; o < 64; o <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)incl, o), z = (uint32_t)__shfl_up((int)lastscan, o);
            if (lane >= o) {
                incl += y;
                lastscan = lastscan == kNone ? z : lastscan;
            }
        }
        uint32_t i = i_base + incl - reps;
        uint32_t cur = (uint32_t)__shfl_up((int)lastscan, 1);
        if (lane == 0 || cur == kNone) cur = carry_last;
        if (first_window) {
            // (the reference looks at the first token's codeword, then at what it repeats)
            const Tok t0 = token(piece, 0);
            const uint32_t sym0 = __builtin_amdgcn_readfirstlane(t0.sym), adv0 = __builtin_amdgcn_readfirstlane(t0.adv);
            if (sym0 == 16) {
                const uint32_t nb0 = adv0 - 2;
                return HeaderResult{b + nb0 > end_b ? RSX_SHORT_INPUT : RSX_BAD_DATA, b};
            }
        }
        // the write pass; the lane in which the last length falls reports where the header ends
        uint32_t end_off = ~0u;
        if (i < total) {
            uint32_t off = start;
            while (off < kSeg && i < total) {
                const Tok t = token(piece, off);
                uint32_t rep = t.rep < total - i ? t.rep : total - i;   // a run passing the end is clamped (src/decompress/mod.rs:462-467)
                if (t.sym < 16) {
                    if (t.sym) L.hdr.lens[i] = (uint8_t)t.sym;
                    cur = t.sym;
                } else if (t.sym == 16) {
                    if (cur)
                        for (uint32_t k = 0; k < rep; k++) L.hdr.lens[i + k] = (uint8_t)cur;
                } else {
                    cur = 0;
                }
                i += rep;
                off += t.adv;
            }
            if (i >= total) end_off = off;
        }
        const unsigned long long done = __ballot(end_off != ~0u);
        if (done) {   // (uniform)
            const uint32_t k = (uint32_t)__ffsll((long long)done) - 1;
            const uint32_t end_pos = b + k * kSeg + (uint32_t)__builtin_amdgcn_readlane((int)end_off, (int)k);
            WAVE_SYNC();
            return HeaderResult{end_pos > end_b ? RSX_SHORT_INPUT : RSX_SUCCESS, end_pos};
        }
        // the header goes on behind this window
        i_base = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) + i_base;
        const uint32_t last63 = (uint32_t)__builtin_amdgcn_readlane((int)lastscan, 63);
        carry_last = last63 == kNone ? carry_last : last63;
        b += 63 * kSeg + (uint32_t)__builtin_amdgcn_readlane((int)exit_off, 63);
    }
}

// src/decompress/mod.rs:355-401: the code lengths of the static code (by the wave)
template <class LS>
__device__ inline void load_static_lens(LS &L)
{
    for (uint32_t i = threadIdx.x & 63; i < 320; i += 64)
        L.hdr.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : i < 288 ? 8 : 5;
    L.hdr.num_litlen = 288;   // (every lane writes the same)
    L.hdr.num_offset = 32;
}

// (The stream's body and its length come by value: a reference to the stream context would keep the
// whole context in scratch memory, written by every lane of every workgroup.)
// First wave's part of a round: when the stream is between blocks, lane 0 reads the next
// block header from the staged window (src/decompress/mod.rs:225-252), then the wave builds
// the decode tables.  Leaves L.in_block / L.stored_len / L.status / L.restage set.
template <class LS>
__device__ __noinline__ void parse_block_header(LS &L, const uint8_t *body, uint64_t body_len, int64_t win_bit0)
{
    constexpr uint32_t kWinLimitBits = LS::kWinLimitBits;
    const uint64_t end_bit = body_len * 8;
    uint32_t b = (uint32_t)((int64_t)L.bitpos - win_bit0);
    L.restage = 0;
    L.hdr.plan = 0;
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
        if (byte + 4 > body_len) {
            L.status = RSX_SHORT_INPUT;
            return;
        }
        const uint8_t *p = body + byte;
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
        L.hdr.plan = 3;   // the wave reads the rest of the header
    } else {
        L.hdr.plan = 1;
    }
    L.hdr.after = b;
}

template <class LS>
__device__ __noinline__ void header_step(LS &L, const uint8_t *body, uint64_t body_len, int64_t win_bit0)
{
    unsigned long long th0 = SPLIT_T_NOW();
    (void)th0;   // (read only by the phase timers of development builds)
    if (threadIdx.x == 0) parse_block_header(L, body, body_len, win_bit0);
    WAVE_SYNC();
    const uint32_t plan = L.hdr.plan;   // (uniform)
    if (plan == 0) return;
    if (plan == 1) {
        load_static_lens(L);
        WAVE_SYNC();
    }
    if (plan == 3) {
        HeaderResult r = read_dynamic_preamble_wave(L, L.hdr.after, win_bit0, body_len * 8);
        if (r.code == RSX_SUCCESS) r = read_code_lengths_wave(L, r.b, win_bit0, body_len * 8);
        WAVE_SYNC();   // (everyone has read hdr.after)
        if (r.code != RSX_SUCCESS) {   // (uniform)
            if (threadIdx.x == 0) L.status = r.code;
            return;
        }
        if (threadIdx.x == 0) L.hdr.after = r.b;
    }
    if (RSX_SPLIT_TIMERS && threadIdx.x == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(13, t - th0); th0 = t; SPLIT_T_ADD(20, 1); }
    const uint32_t num_litlen = L.hdr.num_litlen, num_offset = L.hdr.num_offset;
    // the litlen table first: the offset entries carry the index width of its root
    bool ok = build_decode_table_wave(L.tab + kTabLitlen, kTabLitlen, L.hdr.lens, num_litlen, KIND_LITLEN,
                                      kLitlenBits, 0, L.hdr.sorted_syms, &L.litlen_tablebits);
    WAVE_SYNC();
    ok = ok && build_decode_table_wave(L.tab, 0, L.hdr.lens + num_litlen, num_offset, KIND_OFFSET, kOffsetBits,
                                       L.litlen_tablebits, L.hdr.sorted_syms, nullptr);
    if (RSX_SPLIT_TIMERS && threadIdx.x == 0) SPLIT_T_ADD(14, SPLIT_T_NOW() - th0);
    if (threadIdx.x == 0) {
        if (!ok) {
            L.status = RSX_BAD_DATA;
        } else {
            L.in_block = 1;
            L.bitpos = (uint64_t)(win_bit0 + (int64_t)L.hdr.after);
        }
    }
}

// ---- the symbol decoder ------------------------------------------------------------
// Each lane runs a small state machine whose every step is ONE table lookup: a
// litlen root entry, an offset root entry or a subtable entry.  A literal takes one
// step, a match two (length, distance), long codes one more.  The entry itself says
// which table the next lookup uses, so the step has no state-dependent selects beyond
// "pointer or not", and lanes in different states execute the same instructions: a
// wave that holds literals, match halves and subtable hops at once does not diverge.
//
// Bit window: three consecutive dwords of the staged input per lane (the two that hold the
// next 32 bits and the one behind them, loaded a step ahead), so that a step gets its bits with
// one funnel shift and the LDS round trip of the refill is off the step-to-step dependency chain.
struct Decoder {
    uint32_t w0, w1; // win[b >> 5], win[(b >> 5) + 1]
    uint32_t w2;     // win[(b >> 5) + 2]
    uint32_t b;      // window bit position of the next unread bit
    uint32_t pend;   // value of the last entry that was not a pointer: at a distance, the match length
    uint32_t tbase;  // byte offset of the table of the next lookup
    uint32_t tbits;  // index width of the next lookup

    template <class LS>
    __device__ inline void init(const LS &L, uint32_t at, uint32_t litlen_bits)
    {
        const uint32_t w = at >> 5;
        w0 = L.win[w];
        w1 = L.win[w + 1];
        w2 = L.win[w + 2];
        b = at;
        pend = 0;
        tbase = 4 * kTabLitlen;
        tbits = litlen_bits;
    }
};

// result of one step
struct StepOut {
    uint32_t kind;   // kTokNone (length read / subtable hop), kTokEob, kTokLit, kTokMatch
    uint32_t value;  // literal byte / match distance
    uint32_t len;    // match length when kind == kTokMatch
    uint32_t entry;  // the table entry itself
};

// kValues = false: only the token boundaries matter (literal bytes, lengths and distances are
// not computed)
template <bool kValues = true, class LS>
__device__ inline StepOut decoder_step(const LS &L, Decoder &d)
{
    const uint32_t bits = __builtin_amdgcn_alignbit(d.w1, d.w0, d.b);   // 32 bits from bit b on
    const uint32_t e = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(L.tab) + d.tbase +
                                                          (__builtin_amdgcn_ubfe(bits, 0u, d.tbits) << 2));
    const uint32_t total_bits = e & 31, clen = (e >> 5) & 15;
    const uint32_t field = (e >> 16) & 0x7FFF;
    const bool is_ptr = (int32_t)e < 0;
    // a pointer consumes the root index and has no extra bits; a final entry codeword + extra bits
    const uint32_t value = kValues ? field + __builtin_amdgcn_ubfe(bits >> clen, 0u, total_bits - clen) : 0u;
    // consume: a step takes at most 28 bits, so the window moves by at most one dword
    const uint32_t nb = d.b + total_bits;
    const bool crossed = (nb & 31u) < total_bits;   // the low five bits wrapped
    d.w0 = crossed ? d.w1 : d.w0;
    d.w1 = crossed ? d.w2 : d.w1;
    // unconditional (no branch); first needed after the next crossing
    d.w2 = L.win[(nb >> 5) + 2];
    d.b = nb;

    StepOut o;
    o.kind = __builtin_amdgcn_ubfe(e, 9u, 2u);
    o.value = value;
    o.len = d.pend;
    o.entry = e;
    // next state, from the entry
    d.tbase = is_ptr ? field : (e & kEntNextLit);
    d.tbits = __builtin_amdgcn_ubfe(e, 12u, 4u);
    if (kValues) d.pend = is_ptr ? d.pend : value;
    return o;
}

// What one lane learns from decoding its sub-sequence without emitting.
struct Scan {
    uint32_t exit;   // where the next lane starts (window bit) | kExit* flags
    uint32_t bytes;  // output bytes of the tokens that start in [start, bound)
    uint32_t nm;     // matches among them
};

// Decode tokens that start in [start, bound); end_b = first bit past the body
// (window relative).  Bits past the body read as zero; a step that runs past the
// body stops the chain at the start of its token (ShortInput when it is real).
// kEndFar: the body ends behind everything a step can reach from this window, so no step runs past it
template <bool kEndFar, class LS>
__device__ inline Scan scan_span(const LS &L, uint32_t start, uint32_t bound, int64_t end_b,
                                 uint32_t litlen_bits)
{
    Decoder d;
    d.init(L, start, litlen_bits);
    // window bit positions are far below 2^31: compare in 32 bits
    const uint32_t end32 = end_b < 0 ? 0u : end_b > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)end_b;
    // One exit test per step; everything else is selects.  A step that runs past the body ends
    // the loop, so what it added is taken back behind the loop instead of being guarded inside.
    uint32_t tok = start;   // where the token in flight starts
    uint32_t bytes = 0, nm = 0, inc = 0, kind = kTokNone;
    bool is_match = false, over = false;
    bool running = start < bound;
    while (running) {
        const StepOut o = decoder_step(L, d);
        kind = o.kind;
        over = !kEndFar && d.b > end32;
        is_match = kind == kTokMatch;
        // bit 1 of the kind is 1 for a literal; end of block and unfinished tokens add 0
        inc = is_match ? o.len : __builtin_amdgcn_ubfe(o.entry, 10u, 1u);
        bytes += inc;
        nm += is_match ? 1u : 0u;
        const bool done = kind >= kTokLit;   // a token ended with this step
        tok = (done && !over) ? d.b : tok;
        running = !over && kind != kTokEob && (!done || d.b < bound);
    }
    Scan r;
    r.bytes = over ? bytes - inc : bytes;
    r.nm = (over && is_match) ? nm - 1 : nm;
    // ran past the body: stop at the token start
    r.exit = over ? (tok | kExitStop) : kind == kTokEob ? (d.b | kExitEob) : d.b;
    return r;
}

// A better first guess for a sub-sequence: start kWarmBits in front of it and take the
// first token boundary at or behind `lo`.  Huffman streams re-synchronise, so after a
// stretch of warm-up the decoder is often on real token boundaries already.  An end-of-block
// code does not stop the warm-up: when it is real, the lanes behind it are not part of this
// round anyway, and when it is not, decoding on may still find the boundaries.
template <class LS>
__device__ inline uint32_t warm_start(const LS &L, uint32_t from, uint32_t lo, int64_t end_b,
                                      uint32_t litlen_bits)
{
    Decoder d;
    d.init(L, from, litlen_bits);
    const uint32_t end32 = end_b < 0 ? 0u : end_b > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)end_b;
    // Runs until a token (or a block) ends at or behind lo: every step consumes at least one bit and
    // at most three steps in a row finish nothing (subtable hop, length, subtable hop), so the loop
    // needs no other bound and stays inside the padded window.
    bool running = d.b < lo;
    while (running) {
        const StepOut o = decoder_step<false>(L, d);
        running = o.kind == kTokNone || d.b < lo;
    }
    return d.b <= end32 ? d.b : lo;
}

// exclusive prefix sums of two 32-bit values over the workgroup
template <class LS>
__device__ inline void wg_exclusive_scan2(LS &L, uint32_t a, uint32_t b, uint32_t *ea,
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
    WG_SYNC();
    uint32_t ca = 0, cb = 0;
    for (int k = 0; k < w; k++) {
        ca += L.scan_carry[2 * k];
        cb += L.scan_carry[2 * k + 1];
    }
    *ea = ca + xa - a;
    *eb = cb + xb - b;
    WG_SYNC();
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
    const uint32_t litlen_bits = L.litlen_tablebits;
    const uint32_t round_start = L.round_start_lo;
    // (uniform) the usual case: the body does not end in this window, nor within a token of its limit
    const bool end_far = end_b > (int64_t)kWinLimitBits + 64;
    // lane i owns tokens that start in [b0 + i*S, b0 + (i+1)*S), clipped to the window
    const uint32_t lo = b0 + tid * kSubBits;
    uint32_t bound = lo + kSubBits;
    if (bound > kWinLimitBits) bound = kWinLimitBits;
    // lanes whose sub-sequence lies behind the window take no part in the round (uniform count)
    const uint32_t n_active = (kWinLimitBits - b0 + kSubBits - 1) / kSubBits < (uint32_t)kT
                                  ? (kWinLimitBits - b0 + kSubBits - 1) / kSubBits : (uint32_t)kT;

    // ---- passes 1..k: find every sub-sequence's true start ------------------------
    // A lane that decodes an end-of-block (or runs out of input) while its start is
    // still a guess must not silence the lanes behind it: they keep their own guess
    // until the chain in front of them is settled.
    // Pass 1 decodes every sub-sequence from its guess.  After each pass the owner of
    // sub-sequence i compares its start with where sub-sequence i-1 ended; the ones
    // that changed are compacted into a work list and re-decoded by the first lanes
    // of the workgroup, so late passes (few stragglers) occupy one wave, not all.
    unsigned long long tq = RSX_DEV_TIMERS == 1 ? clock64() : 0;
    // lane 0 starts where the previous round ended; the others warm up in front of their
    // sub-sequence (a lane whose warm-up reaches back to b0 is exact as well)
    uint32_t guess = lo;
    if (tid > 0 && lo < bound) guess = warm_start(L, lo - b0 > kWarmBits ? lo - kWarmBits : b0, lo, end_b, litlen_bits);
    uint32_t start = guess;
    Scan sc;
    if (start >= bound) {
        sc = Scan{start, 0, 0};   // nothing starts here (window exhausted)
    } else {
        sc = end_far ? scan_span<true>(L, start, bound, end_b, litlen_bits)
                     : scan_span<false>(L, start, bound, end_b, litlen_bits);
    }
    L.spec.lane_exit[tid] = sc.exit;
    L.spec.sub_bytes[tid] = (uint16_t)sc.bytes;
    L.spec.sub_nm[tid] = (uint16_t)sc.nm;
    WG_SYNC();
    if (RSX_DEV_TIMERS == 1 && tid == 0) { unsigned long long t = clock64(); L.stat_t[0] += t - tq; tq = t; }
    // the true prefix grows by at least one sub-sequence per pass, so kT passes suffice
    for (int pass = 0; pass <= kT; pass++) {
        WG_SYNC();
        uint32_t ns = guess;
        if (tid > 0) {
            uint32_t pe = L.spec.lane_exit[tid - 1];
            if (!(pe & (kExitEob | kExitStop))) ns = pe;
        }
        const bool need = ns != start && (uint32_t)tid < n_active;
        start = ns;
        // compact the sub-sequences that need another decode
        const unsigned long long bal = __ballot(need);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) L.scan_carry[w] = (uint32_t)__popcll(bal);
        if (need) L.spec.sub_start[tid] = start;
        if (RSX_DEV_TIMERS && tid == 0) L.stat_spec++;
        WG_SYNC();
        uint32_t before = 0, total = 0;
        for (int k = 0; k < kT / 64; k++) {
            uint32_t cnt = L.scan_carry[k];
            if (k < w) before += cnt;
            total += cnt;
        }
        if (total == 0) break;
        if (need) L.spec.work[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = (uint16_t)tid;
        WG_SYNC();
        if ((uint32_t)tid < total) {
            const uint32_t i = L.spec.work[tid];
            const uint32_t st_i = L.spec.sub_start[i];
            uint32_t bound_i = b0 + (i + 1) * kSubBits;
            if (bound_i > kWinLimitBits) bound_i = kWinLimitBits;
            Scan r;
            if (st_i >= bound_i) {
                r = Scan{st_i, 0, 0};   // the previous token overshoots this sub-sequence
            } else {
                r = end_far ? scan_span<true>(L, st_i, bound_i, end_b, litlen_bits)
                            : scan_span<false>(L, st_i, bound_i, end_b, litlen_bits);
            }
            L.spec.lane_exit[i] = r.exit;
            L.spec.sub_bytes[i] = (uint16_t)r.bytes;
            L.spec.sub_nm[i] = (uint16_t)r.nm;
        }
    }
    WG_SYNC();
    sc.exit = L.spec.lane_exit[tid];
    sc.bytes = L.spec.sub_bytes[tid];
    sc.nm = L.spec.sub_nm[tid];
    WG_SYNC();
    if (RSX_DEV_TIMERS == 1 && tid == 0) { unsigned long long t = clock64(); L.stat_t[1] += t - tq; tq = t; }
    // every start now follows from lane 0's: the first lane that ended the block
    // (or the input) is real, and nothing behind it belongs to this block
    if (tid == 0) {
        L.first_flag = kT;
        L.emit_error = ~0u;
    }
    WG_SYNC();
    if (sc.exit & (kExitEob | kExitStop)) atomicMin(&L.first_flag, (uint32_t)tid);
    WG_SYNC();
    const bool live = (uint32_t)tid <= L.first_flag && (uint32_t)tid < n_active;
    if (!live) {
        sc.bytes = 0;
        sc.nm = 0;
    }

    // ---- output positions and the committed prefix ---------------------------------
    uint32_t ex_bytes, ex_nm;
    wg_exclusive_scan2(L, sc.bytes, sc.nm, &ex_bytes, &ex_nm);
    const bool commit = live && ex_bytes + sc.bytes <= kRoundOut && ex_nm + sc.nm <= kMaxMatches;
    // `commit` is monotone over lanes; the last committed lane publishes the round
    L.spec.lane_exit[tid] = commit ? sc.exit | 0x10000000u : sc.exit;
    WG_SYNC();
    const bool last = commit && (tid == kT - 1 || !(L.spec.lane_exit[tid + 1] & 0x10000000u));
    WG_SYNC();   // the match queue shares its storage with the arrays read above

    if (RSX_DEV_TIMERS == 1 && tid == 0) { unsigned long long t = clock64(); L.stat_t[2] += t - tq; tq = t; }
    // ---- emit -------------------------------------------------------------------------
    if (commit) {
        // One exit test per step like the scan; literals and matches are told apart by
        // selects and predicated stores, not branches.  Positions fit 32 bits (slots < 4 GiB).
        uint32_t op = round_start + ex_bytes;
        uint32_t mi = ex_nm;
        int err = -1;
        const uint32_t cap = (uint32_t)c.out_cap;
        const uint32_t end32 = end_b < 0 ? 0u : end_b > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)end_b;
        Decoder d;
        d.init(L, start, litlen_bits);
        auto emit_tokens = [&](auto room_checked) {   // room_checked: neither the slot nor the body can end here
            bool running = d.b < bound;
            while (running) {
                const StepOut o = decoder_step(L, d);
                const bool over = !decltype(room_checked)::value && d.b > end32;
                const bool lit = !over && o.kind == kTokLit, mat = !over && o.kind == kTokMatch;
                const bool eob = !over && o.kind == kTokEob;
                const bool no_room =
                    !decltype(room_checked)::value && ((lit && op >= cap) || (mat && (op > cap || o.len > cap - op)));
                const bool too_far = mat && o.value > op;
                const int e = over ? RSX_SHORT_INPUT : too_far ? RSX_BAD_DATA : no_room ? RSX_INSUFFICIENT_SPACE : -1;
                const bool bad = e >= 0;
                if (lit && !bad) L.ring[(op + c.shift) & kRingMask] = (uint8_t)o.value;
                if (mat && !bad) {
                    L.mq.dst[mi] = (uint16_t)(op - round_start);
                    L.ring[(op + c.shift) & kRingMask] = (uint8_t)(o.value - 1);
                    L.ring[(op + c.shift + 1) & kRingMask] = (uint8_t)((o.value - 1) >> 8);
                    L.ring[(op + c.shift + 2) & kRingMask] = (uint8_t)(o.len - 3);
                }
                // (a failing token ends the loop: what it adds here is never used)
                op += mat ? o.len : lit ? 1u : 0u;
                mi += mat ? 1u : 0u;
                err = bad ? e : err;
                running = !bad && !eob && (o.kind < kTokLit || d.b < bound);
            }
        };
        // a lane whose tokens all end inside the output slot needs no per-token room check, and when
        // the body ends far behind this window no step can run past it; the branch is taken by whole
        // waves (everywhere but at the end of a stream and in a too small slot)
        if (end_far && __all(op + sc.bytes <= cap && op <= cap)) {
            emit_tokens(std::true_type{});
        } else {
            emit_tokens(std::false_type{});
        }
        if (err >= 0) atomicMin(&L.emit_error, ((uint32_t)tid << 8) | (uint32_t)err);
    }
    WG_SYNC();
    if (RSX_DEV_TIMERS == 1 && tid == 0) { unsigned long long t = clock64(); L.stat_t[3] += t - tq; tq = t; }
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
    WG_SYNC();
}

// All lanes: resolve the queued matches of this round, kCopySub output bytes at a
// time.  Every output byte of the step gets a 16-bit pointer to the byte it copies
// (literals point to themselves); pointer jumping (ptr = ptr[ptr]) then makes every
// pointer refer to a byte that is already final -- a literal of this step or data in
// front of it -- in log2(chain length) rounds, whatever the dependency depth
// between matches, overlapping (run-length) matches included.  One gather finishes
// the step.  The pointers live in the window buffer, which is idle during the copy.
template <class LS>
__device__ void copy_round(LS &L, const StreamCtx &c)
{
    constexpr uint32_t kRing = sizeof(L.ring), kRingMask = kRing - 1;   // (this struct's ring, a power of two)
    const uint32_t nm = L.num_matches;
    const uint32_t round_start = L.round_start_lo;
    const uint32_t round_end = L.round_end_lo;
    // positions less than kRing in front of round_end are still in the ring once the round is complete
    const int tid = threadIdx.x;
    uint16_t *ptr = reinterpret_cast<uint16_t *>(L.win);
    // A pointer is the source position minus sub_start, modulo 2^16: [0, kCopySub) points into the
    // step (and may still move), [kLanded, kLanded + kCopySub) into the step at a byte known to be a
    // literal, [2^16 - 32768 - kCopySub, 2^16) in front of the step.  The slot of a literal (nothing to
    // copy) holds the landed pointer to itself, so whoever jumps to it is settled by taking the slot's
    // value like any other.
    constexpr uint32_t kLanded = 0x4000;
    constexpr uint32_t kFront = 0x10000 - 32768 - kCopySub;   // smallest pointer to a byte in front of the step
    static_assert(kLanded >= kCopySub && kLanded + kCopySub < kFront, "pointer ranges are disjoint");
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    uint32_t mcur = 0;                  // first queued match that reaches into the current step (uniform)
    if (tid == 0) ptr[kCopySub] = (uint16_t)kLanded;   // the spare slot (a pad word of the window buffer): read by settled bytes, never used
    // every step starts from a table of self pointers; a step restores it for the next one as soon as
    // its own pointers are in registers
    static_assert(kCopySub == kT * 8, "one 16-byte store per lane restores the pointers");
    // (worked out where it is stored, from a lane number the compiler cannot see through: kept in registers
    // across the steps, these and the eight self pointers of the gather cost the kernel a workgroup per CU)
    auto restore_table = [&](uint32_t t) {
        const uint32_t self0 = (kLanded + 8u * t) * 0x10001u + 0x10000u;   // slots 8 t, 8 t + 1
        reinterpret_cast<uint4 *>(ptr)[t] = make_uint4(self0, self0 + 0x20002u, self0 + 0x40004u, self0 + 0x60006u);
    };
    restore_table((uint32_t)tid);
    WG_SYNC();
    // a match that runs from one step into the next: the second step finds its distance and
    // length here (its first bytes in the ring are data by then); slots alternate between steps
    uint32_t *span = &L.scan_carry[2 * (kT / 64)];   // [2][2] words
    uint32_t step = 0;
    for (uint32_t sub_start = round_start; sub_start < round_end && mcur < nm; sub_start += kCopySub, step++) {
        const uint32_t sub_end = sub_start + kCopySub < round_end ? sub_start + kCopySub : round_end;
        unsigned long long tc = RSX_DEV_TIMERS == 2 ? clock64() : 0;
        unsigned long long ts = SPLIT_T_NOW();
        (void)ts;   // (read only by the phase timers of development builds)
        // matches are sorted by destination: lane t looks at matches mcur + t, + kT, ...
        // The matches that end inside this step are a prefix of them (matches do not overlap);
        // every wave counts its own, the next step starts behind their sum.  The first lane of a
        // wave has the wave's smallest match in every turn of the loop, so it stays in the loop
        // longest and its count is complete.
        auto info_len = [&](uint32_t dst) { return (uint32_t)L.ring[(dst + c.shift + 2) & kRingMask] + 3; };
        uint32_t ended = 0;
        for (uint32_t m = mcur + tid; m < nm; m += kT) {
            const uint32_t dst = round_start + L.mq.dst[m];
            const bool carried = dst < sub_start;   // (only the first match of a step can be)
            uint32_t len, dist;
            if (carried) {
                dist = span[(step & 1) * 2];
                len = span[(step & 1) * 2 + 1];
            } else {
                dist = ((uint32_t)L.ring[(dst + c.shift) & kRingMask] |
                        (uint32_t)L.ring[(dst + c.shift + 1) & kRingMask] << 8) + 1;
                len = info_len(dst);
            }
            const bool ends_here = dst + len <= sub_end;
            ended += (uint32_t)__popcll(__ballot(ends_here));
            if (dst >= sub_end) break;
            if (!ends_here) {
                span[((step + 1) & 1) * 2] = dist;
                span[((step + 1) & 1) * 2 + 1] = len;
            }
            const uint32_t from = dst > sub_start ? dst : sub_start;
            const uint32_t to = dst + len < sub_end ? dst + len : sub_end;
            // four pointers per store: consecutive 16-bit values packed by hand; the source may run
            // from in front of the step into it (0xFFFF -> 0), so the halves are added as 16-bit lanes
            {
                uint32_t at = from - sub_start, left = to - from;
                const uint32_t first = from - dist - sub_start;   // (mod 2^16)
                auto pk = [](uint32_t a, uint32_t b) { return (a & 0xFFFF) | b << 16; };
                auto add2 = [](uint32_t x, uint32_t inc) {
                    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, x) + __builtin_bit_cast(u16x2, inc)));
                };
                uint32_t lo = pk(first, first + 1), hi = pk(first + 2, first + 3);
                for (; left >= 8; left -= 8, at += 8, lo = add2(lo, 0x00080008u), hi = add2(hi, 0x00080008u))
                    *reinterpret_cast<uint4 *>(ptr + at) =
                        make_uint4(lo, hi, add2(lo, 0x00040004u), add2(hi, 0x00040004u));   // (2-byte aligned)
                if (left >= 4) {
                    *reinterpret_cast<uint2 *>(ptr + at) = make_uint2(lo, hi);
                    left -= 4, at += 4, lo = add2(lo, 0x00040004u), hi = add2(hi, 0x00040004u);
                }
                if (left >= 1) ptr[at] = (uint16_t)lo;
                if (left >= 2) ptr[at + 1] = (uint16_t)(lo >> 16);
                if (left >= 3) ptr[at + 2] = (uint16_t)hi;
            }
        }
        if ((tid & 63) == 0) L.scan_carry[tid >> 6] = ended;
        WG_SYNC();
#pragma unroll
        for (int k = 0; k < kT / 64; k++) mcur += L.scan_carry[k];
        if (RSX_DEV_TIMERS == 2 && tid == 0) { unsigned long long t = clock64(); L.stat_t[0] += t - tc; tc = t; }
        if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(15, t - ts); ts = t; SPLIT_T_ADD(19, 1); }
        // Pointer jumping: a pointer into this step takes over the pointer of the byte it points at: a
        // literal's is the landed pointer to that literal, a match byte's is wherever that byte has got
        // to.  Every lane keeps the pointers of its own kPer bytes in registers.  A pointer can still
        // move while it points into the step (v < kCopySub).  The rounds are branch-free: settled bytes
        // read a spare slot behind the table and rewrite their own value.
        constexpr int kPer = kCopySub / kT;
        uint32_t v[kPer];
#pragma unroll
        for (int k = 0; k < kPer; k++) v[k] = ptr[tid + k * kT];   // slots behind the step's end point at themselves
        for (int round = 0; round < 16; round++) {
            uint32_t v2[kPer];
            if (RSX_DEV_TIMERS && tid == 0) L.stat_copy++;
            if (RSX_SPLIT_TIMERS && tid == 0) SPLIT_T_ADD(18, 1);
            // (all reads of a round are issued before the first dependent one is needed)
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                v2[k] = ptr[v[k] < kCopySub ? v[k] : kCopySub];
            }
#if RSX_TWO_BARRIERS
            WG_SYNC();   // every read of this round precedes its writes
#endif
            // (no barrier between the reads and the writes: a lane that reads a pointer its owner has
            // already advanced in this round only gets further along the same chain)
            bool moving = false;
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                const bool pending = v[k] < kCopySub;
                v[k] = pending ? v2[k] : v[k];
                ptr[tid + k * kT] = (uint16_t)v[k];
                moving = moving || v[k] < kCopySub;
            }
            // one barrier publishes the new pointers and the waves' "still moving" flags; odd and even rounds
            // have their own flags: the wave that is first through the barrier writes its next flag while the
            // others still read this round's
            uint32_t *const flags = &L.scan_carry[(round & 1) ? 2 * (kT / 64) + 4 : 4];
            const uint32_t wave_moving = __ballot(moving) != 0;
            if ((tid & 63) == 0) flags[tid >> 6] = wave_moving;
            WG_SYNC();
            uint32_t any_moving = 0;
#pragma unroll
            for (int k = 0; k < kT / 64; k++) any_moving |= flags[k];
            if (!any_moving) break;
        }
        if (RSX_DEV_TIMERS == 2 && tid == 0) { unsigned long long t = clock64(); L.stat_t[2] += t - tc; tc = t; }
        if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(16, t - ts); ts = t; }
        uint32_t lane_no = (uint32_t)tid;
        asm volatile("" : "+v"(lane_no));
        restore_table(lane_no);   // (nobody reads the table any more in this step)
        // gather: sources are literals of this step, earlier bytes of the round (ring) or
        // bytes that have left the ring (read back from the output)
        {
            // sources that have left the ring are read back from the output: every such load of a lane is
            // issued before the first one is waited for (one memory latency per step, not one per byte).
            // A settled pointer v stands for position sub_start + v - 0x10000 (in front of the step) or
            // sub_start + v - kLanded (a literal of the step): both offsets are multiples of the ring's size,
            // so the byte's place in the ring is (sub_start + v) mod kRing for either, and "has left the
            // ring" (more than kRing in front of the round's end, which only a pointer to the front can be)
            // is one range test on v.
            static_assert(kLanded % kRing == 0 && 0x10000 % kRing == 0, "one ring index for both kinds of pointer");
            const uint32_t ring_base = sub_start + c.shift;
            const uint32_t far_span = round_end - sub_start + 0x10000u - kRing - kFront;   // (> 0)
            const __attribute__((address_space(1))) uint8_t *const far_base =
                (const __attribute__((address_space(1))) uint8_t *)c.out + sub_start - 0x10000;
            uint8_t far_byte[kPer], byte[kPer];
            bool far[kPer];
#pragma unroll
            for (int k = 0; k < kPer; k++) far[k] = v[k] - kFront < far_span;
#pragma unroll
            for (int k = 0; k < kPer; k++) {
                far_byte[k] = 0;
                if (far[k]) far_byte[k] = far_base[v[k]];
            }
#pragma unroll
            for (int k = 0; k < kPer; k++) byte[k] = L.ring[(ring_base + v[k]) & kRingMask];   // (an LDS read, never a flat one)
#pragma unroll
            for (int k = 0; k < kPer; k++)
                if (v[k] - (kLanded + k * kT) != lane_no)   // (not a literal, and not a slot behind the step's end)
                    L.ring[(ring_base + lane_no + k * kT) & kRingMask] = far[k] ? far_byte[k] : byte[k];
        }
        WG_SYNC();
        if (RSX_DEV_TIMERS == 2 && tid == 0) { unsigned long long t = clock64(); L.stat_t[3] += t - tc; tc = t; }
        if (RSX_SPLIT_TIMERS && tid == 0) SPLIT_T_ADD(17, SPLIT_T_NOW() - ts);
    }
    WG_SYNC();
}

// All lanes: write output positions [from, to) from the ring to global memory.
// from is on a 16-byte boundary of the global address (or 0); full units use
// 16-byte stores, the ragged first / last unit byte stores.
template <class LS>
__device__ void flush_range(LS &L, const StreamCtx &c, uint32_t from, uint32_t to)
{
    constexpr uint32_t kRingMask = sizeof(L.ring) - 1;
    if (to <= from) return;   // (uniform)
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

// zlib / gzip header checks, src/decompress/mod.rs:1079-1095 and :1149-1192.  With kFormatVerifyFhcrc in
// `format` the header CRC of a gzip member that has one (FLG.FHCRC) is checked as well: the low 16 bits of the
// CRC-32 of the header bytes in front of it (RFC 1952 2.3.1); the reference skips the field (:1186-1188).
__device__ ContainerInfo parse_container(int format, const uint8_t *in, uint64_t in_len)
{
    const bool verify_fhcrc = (format & kFormatVerifyFhcrc) != 0;
    format &= kFormatMask;
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
        if (verify_fhcrc && (flg & 0x02)) {
            uint32_t crc = ~0u;   // (headers are tens of bytes: bit by bit, no table)
            for (uint64_t k = 0; k + 2 < h; k++) {
                crc ^= in[k];
                for (int b = 0; b < 8; b++) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1)));
            }
            if (((~crc) & 0xFFFF) != ((uint32_t)in[h - 2] | ((uint32_t)in[h - 1] << 8))) {
                ci.status = RSX_BAD_DATA;
                return ci;
            }
        }
        ci.hdr = (uint32_t)h;
        ci.trailer = 8;
    }
    return ci;
}

// Decoding that can be taken up again (rsx_inflate_stream_step, the engine's side of
// Decompressor::decompress_streaming, src/decompress/mod.rs:213-353).  An item then starts at bit
// start_bit[i] of its first byte, the first hist_len[i] bytes of its output slot are the output produced so far
// (at most the 32 KiB a match can reach back), and the decoder reports the last place BETWEEN TWO BLOCKS it
// reached: a stream whose input or output room ends inside a block is reported up to the end of the block
// before (status RSX_SHORT_INPUT / RSX_INSUFFICIENT_SPACE, out_sizes = the bytes behind the history up to there,
// end_bits = the item's bits up to there), so that the caller can go on from that bit with more input or room.
// Damaged data reports the blocks in front of the damaged one the same way, with RSX_BAD_DATA.
struct ResumeArgs {
    const uint32_t *start_bit;   // null: items are whole streams (every other field is unused)
    const uint64_t *hist_len;
    uint64_t *end_bits;
};

// STATS builds add per-stream phase timers (development aid, RSX_INFLATE_STATS=1):
// stats[item*8 + k]: 0 rounds, 1 speculation passes, 2 copy passes, 3..7 cycles in
// stage / header / decode / copy / flush
template <bool STATS>
__global__ __launch_bounds__(kT, 5) void inflate_kernel(
    int format, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    uint64_t total_in, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status, uint64_t *__restrict__ body_end,
    unsigned long long *__restrict__ stats, int only_status, const uint64_t *__restrict__ in_hi,
    const uint64_t *__restrict__ out_hi, const uint64_t *__restrict__ in_base_p, uint32_t *__restrict__ taken,
    ResumeArgs resume)
{
    unsigned long long tS = 0, tH = 0, tD = 0, tC = 0, tF = 0, t0 = 0, nRounds = 0;
#define STAT_BEGIN() do { if (STATS) t0 = clock64(); } while (0)
#define STAT_END(acc) do { if (STATS) acc += clock64() - t0; } while (0)
    __shared__ StreamLds L;
    const uint32_t item = blockIdx.x;
    if (item >= n) return;
    // only_status != 0: this launch only takes the streams another kernel has marked with it (uniform)
    if (only_status && status[item] != only_status) return;
    const int tid = threadIdx.x;
    if (taken && tid == 0) atomicAdd(taken, 1u);   // (streams this launch decodes: the ones the two kernels left)

    // item i is in[in_off[i], in_hi[i]) -> out[out_off[i], out_hi[i]); in_hi = in_off + 1 and
    // out_hi = out_off + 1 for the usual prefix arrays
    const uint64_t ib = in_off[item], ie = in_hi[item];
    const uint64_t ob = out_off[item], oe = out_hi[item];
    StreamCtx c;
    c.in_alloc = in + *in_base_p;   // the batch's bytes are in[base .. base + total_in)
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

    const bool resumable = resume.start_bit != nullptr;   // (uniform)
    // history: output positions below `hist` were produced by earlier calls
    uint32_t hist = 0;
    if (resumable) {
        const uint64_t h = resume.hist_len[item];
        hist = (uint32_t)(h < c.out_cap ? h : c.out_cap);
    }
    if (tid == 0) {
        L.bitpos = resumable ? (resume.start_bit[item] & 7u) : 0;
        L.status = -1;
        L.in_block = 0;
        L.final_block = 0;
        L.stored_len = -1;
        L.restage = 0;
        L.emit_error = ~0u;
        L.stat_spec = 0;
        L.stat_copy = 0;
        L.stat_t[0] = L.stat_t[1] = L.stat_t[2] = L.stat_t[3] = 0;
        L.round_start_lo = hist;
        L.round_end_lo = hist;
        L.num_matches = 0;
    }
    // the ring mirrors the newest output bytes: what it would hold had this call produced the history
    for (uint32_t k = tid; k < kRing && k < hist; k += kT) {
        const uint32_t pos = hist - 1 - k;
        L.ring[(pos + c.shift) & kRingMask] = c.out[pos];
    }
    uint32_t flushed = hist;  // output positions below this are in global memory (uniform)
    uint64_t ck_bit = resumable ? (resume.start_bit[item] & 7u) : 0;   // the last place between two blocks (uniform)
    uint32_t ck_out = hist;
    uint64_t seen_bitpos = ~0ull;  // progress watchdog (uniform)
    int stalled = 0;
    WG_SYNC();

    for (;;) {
        // every round either consumes input or moves the window; anything else is a bug
        // and must not spin on the GPU
        {
            const uint64_t bp = L.bitpos;
            if (!L.in_block && L.stored_len < 0) {   // (uniform) between two blocks
                ck_bit = bp;
                ck_out = L.round_end_lo;
            }
            stalled = bp == seen_bitpos ? stalled + 1 : 0;
            seen_bitpos = bp;
            WG_SYNC();
            if (stalled > 4) {
                if (tid == 0) L.status = RSX_INTERNAL_ERROR;   // (not the data's fault: its own code)
                WG_SYNC();
                break;
            }
        }
        int64_t win_bit0;
        STAT_BEGIN();
        stage_window(L, c, &win_bit0);
        WG_SYNC();
        STAT_END(tS);
        nRounds++;
        if (!L.in_block) {
            STAT_BEGIN();
            WG_SYNC();
            if (tid < 64) header_step(*(LdsStream *)&L, c.body, c.body_len, win_bit0);
            WG_SYNC();
            STAT_END(tH);
        }
        {
            // snapshot the control block, then let nobody change it before all have read it
            const int restage = L.restage, st0 = L.status, stored0 = L.stored_len;
            const uint32_t b0 = (uint32_t)((int64_t)L.bitpos - win_bit0);
            WG_SYNC();
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
        WG_SYNC();
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
                    WG_SYNC();
                    pos += piece;
                    done += piece;
                    uint32_t up2 = ((pos + c.shift) & ~15u);
                    up2 = up2 > c.shift ? up2 - c.shift : 0;
                    if (up2 < flushed) up2 = flushed;
                    flush_range(L, c, flushed, up2);
                    flushed = up2;
                    WG_SYNC();
                }
                if (L.final_block) st = RSX_SUCCESS;
            }
            WG_SYNC();
            if (tid == 0) {
                L.stored_len = -1;
                L.bitpos += (uint64_t)len * 8;
                L.round_start_lo = pos;
                L.round_end_lo = pos;
                L.status = st;
            }
            WG_SYNC();
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
        WG_SYNC();
        STAT_END(tF);
    }

    const int st = L.status;
    const uint32_t produced = L.round_end_lo;
    if (st == RSX_SUCCESS) flush_range(L, c, flushed, produced);
    if (resumable && st != RSX_SUCCESS && st != RSX_INTERNAL_ERROR) {
        // up to the end of the last whole block: what is left of it in the ring (the tail of the round that
        // ended it: rounds behind it wrote past it, and a failing round copies nothing)
        if (ck_out > flushed) flush_range(L, c, flushed, ck_out);
    }
    if (tid == 0) {
        status[item] = st;
        out_sizes[item] = st == RSX_SUCCESS ? produced : 0;
        if (resumable) {
            const bool partial = st != RSX_SUCCESS && st != RSX_INTERNAL_ERROR;
            out_sizes[item] = st == RSX_SUCCESS ? produced - hist : partial ? ck_out - hist : 0;
            resume.end_bits[item] = st == RSX_SUCCESS ? L.bitpos : partial ? ck_bit : 0;
        }
        // first byte behind the DEFLATE stream, relative to the item start
        body_end[item] = ci.hdr + ((L.bitpos + 7) >> 3);
        if (STATS) {
            unsigned long long *q = stats + (size_t)item * 12;
            q[0] = nRounds; q[1] = L.stat_spec; q[2] = L.stat_copy;
            q[3] = L.stat_t[0]; q[4] = tH; q[5] = tD; q[6] = tC; q[7] = L.stat_t[1];
            q[8] = L.stat_t[2]; q[9] = L.stat_t[3]; q[10] = tS; q[11] = tF;
        }
    }
#undef STAT_BEGIN
#undef STAT_END
}

// =====================================================================================
// The two-kernel decoder: inflate_decode_kernel turns the Huffman-coded body of every
// stream into an array of LZ77 tokens in device memory, inflate_expand_kernel turns the
// tokens into bytes.  Splitting the work frees each half from the other's LDS: the first
// kernel has no output ring, so one round can hold 256 sub-sequences of up to 504 bits (a
// 16 KiB window) and the warm-up in front of a sub-sequence is no longer several times its
// length; the second has no decode tables and no serial decoding at all.  Both leave every
// anomaly (a damaged stream, an output slot that is too small, a stream that needs more
// token space than the workspace has) to inflate_kernel above, which then decodes that
// stream again from its start and reports the reference's status code: the two kernels
// only ever finish streams that decode cleanly.
//
// Token array of a stream: rows of 256 words.  A round of the first kernel owns a header
// row (word l: tokens | matches << 8 | output bytes << 16 of sub-sequence l), kSegRows rows of
// running totals (row p: output bytes | matches << 16 of the first 16 p tokens of every sub-sequence,
// so that the second kernel can start in the middle of one), and then
// as many rows as its longest sub-sequence has tokens; row k holds the k-th token of
// every sub-sequence: a literal's byte, or a match (1 << 31 | length - 3 << 15 | distance - 1).
// A decoder step that finishes no token (a length still waiting for its distance, a subtable hop,
// the end of the block) writes nothing.  A row per token of every lane keeps the stores of a wave
// close together (lanes that are a token apart write neighbouring rows).  Rounds are listed in a
// per-stream array of RoundRec.

constexpr int32_t kStatusPending = -2;    // decoded into tokens, not expanded yet
constexpr int32_t kStatusFallback = -3;   // left to inflate_kernel
constexpr int32_t kStatusChain = -5;      // a part that copies from the part in front of it / a stream with such parts
// The first kernel runs kDT threads per stream: a round has at most kDT sub-sequences, each a column of its token
// rows (rows are kT words wide whatever kDT is: rounds share them side by side).  Fewer, longer sub-sequences
// spend a smaller share of the decoder's steps on the warm-up in front of a sub-sequence.
#ifndef RSX_DEC_THREADS
#define RSX_DEC_THREADS 256
#endif
constexpr int kDT = RSX_DEC_THREADS;
static_assert(kDT == 256 || kDT == 128, "whole waves, and the rows hold a whole number of rounds side by side");
constexpr uint32_t kTokCap = 96 * (kT / kDT);   // tokens per sub-sequence and round
constexpr uint32_t kSegSteps = 16;        // the expander walks a sub-sequence in pieces of this many steps
constexpr uint32_t kSegRows = kTokCap / kSegSteps - 1;   // header rows 1..kSegRows: bytes | matches << 16 in front of piece p
constexpr uint32_t kHdrRows = 1 + kSegRows;
static_assert(kTokCap % kSegSteps == 0 && (kSegSteps & (kSegSteps - 1)) == 0, "whole pieces, and the piece of a token is found by masking");
#ifndef RSX_LANE_BYTES
#define RSX_LANE_BYTES 8192
#endif
constexpr uint32_t kLaneBytesStop = RSX_LANE_BYTES; // a sub-sequence stops once it has produced this many bytes
static_assert(kLaneBytesStop + 258 <= kRoundOut, "one sub-sequence must fit a round of the expander");
static_assert(kTokCap <= kMaxMatches, "one sub-sequence must fit the match queue");
// Sub-sequence lengths are odd multiples of 32 bits: lane i starts at word i * length / 32 of the window, and
// with an odd stride the 64 lanes of a wave start in 64 different LDS banks (448 bits = 14 words: two lanes
// per bank, 4.69 ms for the bench's decode kernel; 480 bits: 4.32 ms; 512 bits, 16 lanes per bank: 5.02 ms).
#ifndef RSX_DEC_SUB_BITS
#define RSX_DEC_SUB_BITS (RSX_DEC_THREADS == 256 ? 480 : 992)
#endif
#ifndef RSX_DEC_WARM_BITS
#define RSX_DEC_WARM_BITS 1024
#endif
constexpr uint32_t kDecSubMax = RSX_DEC_SUB_BITS;      // longest sub-sequence (bits)
constexpr uint32_t kDecSubMin = 32;
constexpr uint32_t kDecWarm = RSX_DEC_WARM_BITS;
constexpr uint32_t kDecWinBytes = (kDT * kDecSubMax / 8 + 256 + 15) / 16 * 16;   // the window of one round
constexpr uint32_t kExitFull = 1u << 29;  // the sub-sequence stopped early: token or byte limit

enum { ROUND_END = 0, ROUND_TOKENS = 1, ROUND_STORED = 2 };
struct RoundRec {
    uint32_t kind;      // ROUND_*
    uint32_t a;         // tokens: first row (the header row);   stored: length in bytes
    uint32_t b;         // tokens: number of sub-sequences | first column << 16;   stored: low half of the source offset in the body
    uint32_t c;         // tokens: 0, or 1 + the pool row its chunk starts at (a is then relative to the chunk);   stored: high half
};

// A stream owns one token row per 512 bits of its input (and the longest round on top).  Text whose blocks
// are a megabyte long needs a row per 850 bits; every block end cuts a round short, so streams of 64 KiB
// blocks need one per 550, and one per 1024 sent a third of them to the one-kernel decoder.
constexpr uint32_t kRowBitsLog2 = 9;
// where a stream's token rows and round records live (both kernels compute the same)
constexpr uint32_t kPoolChunk = 256;   // rows a stream takes from the shared pool at a time
struct SplitLayout {
    uint32_t *tok;          // token rows of all streams
    RoundRec *rounds;       // round records of all streams
    // Rows behind the streams' own, shared by the batch: a stream that has used up its share (blocks so short
    // that most rounds fill a few columns of their rows, streams of nothing but literals) takes kPoolChunk rows
    // at a time from here, first come first served; its round records then say which chunk a round's rows are in.
    uint64_t pool_first;    // first row of the pool
    uint32_t pool_rows;     // rows in it
    unsigned long long *pool_next;   // rows asked for so far (cleared per call; 64 bits: every stream of a batch may ask)
    // rel = the stream's input offset minus the batch's first
    __device__ uint64_t first_row(uint64_t rel, uint32_t item) const
    {
        return (rel * 8 >> kRowBitsLog2) + (uint64_t)item * (kTokCap + kHdrRows + 1);
    }
    __device__ uint64_t row_budget(uint64_t in_len) const { return (in_len * 8 >> kRowBitsLog2) + kTokCap + kHdrRows; }
    // (a round per block at least: zlib at memLevel 1 writes a block every 127 symbols, 27 bytes of input when
    // the data is one repeated byte and the strategy Z_HUFFMAN_ONLY)
    __device__ uint64_t first_round(uint64_t rel, uint32_t item) const { return (rel >> 4) + (uint64_t)item * 66; }
    __device__ uint64_t round_budget(uint64_t in_len) const { return (in_len >> 4) + 64; }
};
static inline uint64_t split_tok_rows(size_t n, uint64_t total_in) { return (total_in * 8 >> kRowBitsLog2) + n * (kTokCap + kHdrRows + 1) + 2; }
static inline uint64_t split_pool_rows(uint64_t total_in)
{
    const uint64_t r = (total_in * 8 >> kRowBitsLog2) / 4;   // a quarter on top of the streams' own rows
    return r < 1024 ? 1024 : r > 0x7FFFFF00ull ? 0x7FFFFF00ull : r;
}
static inline uint64_t split_round_recs(size_t n, uint64_t total_in) { return (total_in >> 4) + n * 66 + 2; }

typedef StreamLdsT<kDecWinBytes, 16> DecodeLds;
typedef RSX_LDS DecodeLds LdsDecode;

struct TokScan {
    uint32_t exit;   // where the next sub-sequence starts (window bit) | kExit* flags
    uint32_t info;   // tokens | matches << 8 | bytes << 16
};

// scan_span that also writes the tokens it decodes: hdr = this sub-sequence's word of the round's first
// header row; what step k finished goes to hdr[(kHdrRows + k) * kT], the totals in front of piece p to hdr[p * kT]
template <bool kEndFar, class LS>
__device__ inline TokScan scan_tokens(const LS &L, uint32_t start, uint32_t bound, int64_t end_b,
                                      uint32_t litlen_bits, uint32_t *__restrict__ hdr)
{
    uint32_t *__restrict__ col = hdr + kHdrRows * kT;
    Decoder d;
    d.init(L, start, litlen_bits);
    const uint32_t end32 = end_b < 0 ? 0u : end_b > 0x7FFFFFFF ? 0x7FFFFFFFu : (uint32_t)end_b;
    uint32_t tok = start;   // where the token in flight starts
    uint32_t bytes = 0, nm = 0, kind = kTokNone;
    uint32_t ntok = 0;   // tokens written: a step that finishes none (a length, a subtable hop) writes nothing
    bool over = false, full = false;
    bool running = start < bound;
    while (running) {
        const StepOut o = decoder_step(L, d);
        kind = o.kind;
        over = !kEndFar && d.b > end32;
        const bool is_match = kind == kTokMatch;
        const bool fin = kind >= kTokLit && !over;   // a literal or a match ends with this step, inside the body
        const uint32_t rec = is_match ? (0x80000000u | ((o.len - 3) << 15) | (o.value - 1)) : o.value;
        if (fin) col[ntok * kT] = rec;
        ntok += fin ? 1u : 0u;
        bytes += fin ? (is_match ? o.len : 1u) : 0u;
        nm += (fin && is_match) ? 1u : 0u;
        if (fin && (ntok & (kSegSteps - 1)) == 0) hdr[(ntok / kSegSteps) * kT] = bytes | (nm << 16);
        tok = fin ? d.b : tok;
        // (a sub-sequence stops early only at the end of a token)
        full = fin && (ntok + 4 >= kTokCap || bytes >= kLaneBytesStop);
        running = !over && kind != kTokEob && !full && (kind < kTokLit || d.b < bound);
    }
    TokScan r;
    r.info = ntok | (nm << 8) | (bytes << 16);
    r.exit = over ? (tok | kExitStop) : kind == kTokEob ? (d.b | kExitEob) : (full && d.b < bound) ? (d.b | kExitFull) : d.b;
    return r;
}

// One round of the first kernel: all lanes decode the Huffman block body in the staged window from
// L.bitpos on, sub-sequences of `sub_bits` bits, and leave the tokens in this round's rows.  Returns
// (uniform) the number of rows the round used, header row included.  *sub_next receives the sub-sequence
// length to use for the next round, *nlanes_ret the number of sub-sequences in the round.
// `rows` points at the round's first column of its rows; the round has at most `max_lanes` sub-sequences (the
// columns that are left of the rows when rounds share them).
__device__ uint32_t decode_round_tokens(DecodeLds &L, const StreamCtx &c, int64_t win_bit0, uint32_t sub_bits,
                                        uint32_t *__restrict__ rows, uint32_t max_lanes, uint32_t *nlanes_ret,
                                        uint32_t *sub_next, uint32_t *bytes_ret)
{
    constexpr uint32_t kWinLimitBits = DecodeLds::kWinLimitBits;
    const int tid = threadIdx.x;
    const uint32_t b0 = (uint32_t)((int64_t)L.bitpos - win_bit0);
    const int64_t end_b = (int64_t)(c.body_len * 8) - win_bit0;  // first bit past the body
    const uint32_t litlen_bits = L.litlen_tablebits;
    const bool end_far = end_b > (int64_t)kWinLimitBits + 64;
    const uint32_t lo = b0 + tid * sub_bits;
    uint32_t bound = lo + sub_bits;
    if (bound > kWinLimitBits) bound = kWinLimitBits;
    uint32_t n_active = (kWinLimitBits - b0 + sub_bits - 1) / sub_bits;
    if (n_active > max_lanes) n_active = max_lanes;
    if (n_active > (uint32_t)kDT) n_active = kDT;
    if ((uint32_t)tid >= n_active) bound = lo;   // (a lane without a column decodes nothing)
    uint32_t *col = rows + tid;        // this lane's word of the first header row

    uint32_t guess = lo;
    unsigned long long tq = SPLIT_T_NOW();
    (void)tq;   // (read only by the phase timers of development builds)
    if (tid > 0 && lo < bound) guess = warm_start(L, lo - b0 > kDecWarm ? lo - kDecWarm : b0, lo, end_b, litlen_bits);
    if (RSX_SPLIT_TIMERS && tid == 64) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(2, t - tq); tq = t; }
    uint32_t start = guess;
    TokScan sc;
    if (start >= bound) {
        sc = TokScan{start, 0};
    } else {
        sc = end_far ? scan_tokens<true>(L, start, bound, end_b, litlen_bits, col)
                     : scan_tokens<false>(L, start, bound, end_b, litlen_bits, col);
    }
    L.tspec.lane_exit[tid] = sc.exit;
    L.tspec.info[tid] = sc.info;
    WG_SYNC();
    if (RSX_SPLIT_TIMERS && tid == 64) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(3, t - tq); tq = t; }
    // passes until every sub-sequence starts where its predecessor ended (as decode_round_parallel)
    for (int pass = 0; pass <= kDT; pass++) {
        WG_SYNC();
        uint32_t ns = guess;
        if (tid > 0) {
            const uint32_t pe = L.tspec.lane_exit[tid - 1];
            if (!(pe & (kExitEob | kExitStop | kExitFull))) ns = pe;
        }
        const bool need = ns != start && (uint32_t)tid < n_active;
        start = ns;
        const unsigned long long bal = __ballot(need);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) L.scan_carry[w] = (uint32_t)__popcll(bal);
        if (need) L.tspec.sub_start[tid] = start;
        WG_SYNC();
        uint32_t before = 0, total = 0;
        for (int k = 0; k < kDT / 64; k++) {
            const uint32_t cnt = L.scan_carry[k];
            if (k < w) before += cnt;
            total += cnt;
        }
        if (total == 0) break;
        if (RSX_SPLIT_TIMERS && tid == 64) SPLIT_T_ADD(7, 1);
        if (need) L.tspec.work[before + (uint32_t)__popcll(bal & ((1ull << lane) - 1))] = (uint16_t)tid;
        // A sub-sequence that is decoded again is decoded by another thread, often of another wave, and its
        // tokens go to the same words of global memory: the stores of the pass before must have arrived
        // before this pass's are sent, or an old token could land on a new one (the barrier orders the
        // waves' instructions, not their stores in flight).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WG_SYNC();
        if ((uint32_t)tid < total) {
            const uint32_t i = L.tspec.work[tid];
            const uint32_t st_i = L.tspec.sub_start[i];
            uint32_t bound_i = b0 + (i + 1) * sub_bits;
            if (bound_i > kWinLimitBits) bound_i = kWinLimitBits;
            TokScan r;
            if (st_i >= bound_i) {
                r = TokScan{st_i, 0};
            } else {
                r = end_far ? scan_tokens<true>(L, st_i, bound_i, end_b, litlen_bits, rows + i)
                            : scan_tokens<false>(L, st_i, bound_i, end_b, litlen_bits, rows + i);
            }
            L.tspec.lane_exit[i] = r.exit;
            L.tspec.info[i] = r.info;
        }
    }
    WG_SYNC();
    if (RSX_SPLIT_TIMERS && tid == 64) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(4, t - tq); tq = t; }
    sc.exit = L.tspec.lane_exit[tid];
    sc.info = L.tspec.info[tid];
    if (tid == 0) L.first_flag = n_active - 1;
    WG_SYNC();
    if ((sc.exit & (kExitEob | kExitStop | kExitFull)) && (uint32_t)tid < n_active) atomicMin(&L.first_flag, (uint32_t)tid);
    WG_SYNC();
    const uint32_t last = L.first_flag;   // the round ends with this sub-sequence
    const bool live = (uint32_t)tid <= last;
    // header row, the longest column, totals for the next round's sub-sequence length
    if (live) rows[tid] = sc.info;
    uint32_t mx = live ? (sc.info & 0xFF) : 0u, tk = mx, by = live ? (sc.info >> 16) : 0u;
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (uint32_t)__shfl_xor(mx, o));
        tk += __shfl_xor(tk, o);
        by += __shfl_xor(by, o);
    }
    if ((tid & 63) == 0) {
        L.scan_carry[tid >> 6] = mx;
        L.scan_carry[4 + (tid >> 6)] = tk;
        L.scan_carry[8 + (tid >> 6)] = by;
    }
    WG_SYNC();
    mx = tk = by = 0;
#pragma unroll
    for (int k = 0; k < kDT / 64; k++) {
        mx = max(mx, L.scan_carry[k]);
        tk += L.scan_carry[4 + k];
        by += L.scan_carry[8 + k];
    }
    const uint32_t last_exit = L.tspec.lane_exit[last];
    WG_SYNC();
    const uint32_t consumed = (last_exit & kExitPosMask) - b0;
    // next round: aim at 5/8 of the token limit and 3/4 of the byte limit per sub-sequence
    {
        uint32_t s = kDecSubMax;
        if (tk) s = min(s, (uint32_t)((uint64_t)consumed * (kTokCap * 5 / 8) / tk));   // (tk: tokens)
        if (by) s = min(s, (uint32_t)((uint64_t)consumed * (kLaneBytesStop * 3 / 4) / by));
        if (last_exit & kExitFull) s = min(s, sub_bits / 2);
        s = max(s, kDecSubMin);
        *sub_next = (((s >> 5) - 1) | 1u) << 5;   // the largest odd multiple of 32 that is not above it
    }
    // (bit 31: the round ends with a sub-sequence that ran into the token or byte limit)
    *nlanes_ret = (last + 1) | ((last_exit & kExitFull) ? 0x80000000u : 0u);
    *bytes_ret = by;
    if (tid == 0) {
        L.bitpos = (uint64_t)(win_bit0 + (int64_t)(last_exit & kExitPosMask));
        if (last_exit & kExitStop) {
            L.status = RSX_SHORT_INPUT;   // (any failure sends the stream to inflate_kernel)
        } else if (last_exit & kExitEob) {
            L.in_block = 0;
            if (L.final_block) L.status = RSX_SUCCESS;
        }
    }
    // The sub-sequences behind the one that ends the round have stored tokens too, into rows that the next
    // round takes for its own: their stores must have arrived before the next round's are sent (by then
    // they have had the passes and the round's tail to do so: the wait costs nothing, but without it an old
    // token could land on a new one).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_SYNC();
    if (RSX_SPLIT_TIMERS && tid == 64) { SPLIT_T_ADD(5, SPLIT_T_NOW() - tq); SPLIT_T_ADD(6, 1); }
    return kHdrRows + mx;
}

#ifndef RSX_DEC_OCC
#define RSX_DEC_OCC (RSX_DEC_THREADS == 256 ? 6 : 3)
#endif
// parts != nullptr: the items are the parts of cut streams (see "Streams cut into parts" below): *parts->count
// of them, the ones marked kStatusFallback beforehand are left alone, a part with open_end set ends well when
// its input ends between two blocks, and every part reports the bytes its tokens produce.
struct DecodeParts {
    const uint32_t *count;
    const uint8_t *open_end;
    uint64_t *bytes;
};
__global__ __launch_bounds__(kDT, RSX_DEC_OCC) void inflate_decode_kernel(
    int format, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    uint64_t total_in, int32_t *__restrict__ status, uint64_t *__restrict__ body_end, SplitLayout lay,
    const uint64_t *__restrict__ in_hi, const uint64_t *__restrict__ in_base_p, DecodeParts parts)
{
    __shared__ DecodeLds L;
    const uint32_t item = blockIdx.x;
    if (item >= n) return;
    if (parts.count && (item >= *parts.count || status[item] == kStatusFallback)) return;   // (uniform)
    const bool open_end = parts.open_end && parts.open_end[item];
    uint64_t produced = 0;   // output bytes of the tokens and stored blocks so far (uniform)
    const int tid = threadIdx.x;
    const uint64_t ib = in_off[item], ie = in_hi[item];
    StreamCtx c;
    const uint64_t in_base = *in_base_p;
    c.in_alloc = in + in_base;
    c.in_alloc_len = total_in;
    c.out = nullptr;
    c.out_cap = 0;
    c.shift = 0;
    // (an item that does not lie inside the batch's bytes would put its token rows anywhere)
    const bool inside = ib >= in_base && ie >= ib && ie - in_base <= total_in;
    const ContainerInfo ci = inside ? parse_container(format, in + ib, ie - ib) : ContainerInfo{0, 0, RSX_BAD_DATA};
    if (ci.status >= 0) {   // (uniform) inflate_kernel reports it
        if (tid == 0) status[item] = kStatusFallback;
        return;
    }
    c.body = in + ib + ci.hdr;
    c.body_len = (ie - ib) - ci.hdr - ci.trailer;
    uint32_t *const tok = lay.tok + lay.first_row(ib - in_base, item) * kT;   // (the stream's own rows)
    RoundRec *const rounds = lay.rounds + lay.first_round(ib - in_base, item);
    const uint64_t own_budget = lay.row_budget(ie - ib), round_budget = lay.round_budget(ie - ib);
    uint64_t row = 0, nrounds = 0;   // (uniform) row: rows used of the stream's own, or of its current chunk of the pool
    // Rounds share rows: a round ends where its block ends, and one that filled k of the 256 columns of its rows
    // leaves the others to the rounds behind it, as long as enough of them are left.  Streams of short blocks
    // (zlib with a small memLevel, a flush every few KB) would otherwise own rows that are mostly empty.
    // row = first row of the open set of rows, set_rows = its height so far, col0 = its first free column.
    uint32_t set_rows = 0, col0 = 0;

    if (tid == 0) {
        L.bitpos = 0;
        L.status = -1;
        L.in_block = 0;
        L.final_block = 0;
        L.stored_len = -1;
        L.restage = 0;
        L.pool_chunk = 0;
    }
    uint32_t sub_bits = kDecSubMax;
    uint64_t seen_bitpos = ~0ull;
    int stalled = 0;
    bool bail = false;
    SPLIT_T_BEGIN();
    WG_SYNC();
    for (;;) {
        {
            const uint64_t bp = L.bitpos;
            const bool between = !L.in_block && L.stored_len < 0;
            stalled = bp == seen_bitpos ? stalled + 1 : 0;
            seen_bitpos = bp;
            WG_SYNC();
            if (stalled > 4) {
                bail = true;
                break;
            }
            // a part of a cut stream ends where the next one starts: between two blocks, at the last bit of
            // its input
            if (open_end && between && bp != 0 && bp == c.body_len * 8) {
                if (tid == 0) L.status = RSX_SUCCESS;
                WG_SYNC();
                break;
            }
        }
        // room for one more round record and for the end record behind it
        if (nrounds + 2 > round_budget) {
            bail = true;
            break;
        }
        int64_t win_bit0;
        unsigned long long ts = SPLIT_T_NOW();
        (void)ts;   // (read only by the phase timers of development builds)
        stage_window<DecodeLds, kDT>(L, c, &win_bit0);
        WG_SYNC();
        if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(0, t - ts); ts = t; }
        if (!L.in_block) {
            WG_SYNC();
            if (tid < 64) header_step(*(LdsDecode *)&L, c.body, c.body_len, win_bit0);
            WG_SYNC();
            if (RSX_SPLIT_TIMERS && tid == 0) SPLIT_T_ADD(1, SPLIT_T_NOW() - ts);
        }
        const int restage = L.restage, st0 = L.status, stored0 = L.stored_len;
        const uint32_t b0 = (uint32_t)((int64_t)L.bitpos - win_bit0);
        // 0: the stream writes to its own rows; else 1 + the pool row its current chunk starts at (kept in LDS: the
        // kernel has no register to spare, and it is asked for once per round)
        uint32_t chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.pool_chunk);
        WG_SYNC();
        if (restage) {
            if (tid == 0) L.restage = 0;
            continue;
        }
        if (st0 < 0 && stored0 < 0) {
            if (b0 + sub_bits + 64 > DecodeLds::kWinLimitBits) continue;
            uint64_t row_budget = chunk ? (uint64_t)kPoolChunk : own_budget;
            // room for the longest round of tokens (asked only where tokens follow: a stored block needs no rows).
            // A token is at least one bit long and starts inside its sub-sequence: short sub-sequences (the
            // length adapts to streams of few, long matches) need few rows, and the streams that have them
            // are the short ones that own few.
            // A stream that runs short of rows goes on with shorter sub-sequences (a round then needs few rows
            // at worst, and uses fewer).
            while (row + kHdrRows + min(kTokCap, sub_bits + 1) > row_budget && sub_bits > kDecSubMin)
                sub_bits = max(kDecSubMin, (((sub_bits >> 6) - 1) | 1u) << 5);   // about half, an odd multiple of 32 bits
            if (row + kHdrRows + min(kTokCap, sub_bits + 1) > row_budget) {
                // out of rows: on with a chunk of the pool, as long as it lasts
                // (everyone has read L.pool_chunk in front of the barrier above)
                if (tid == 0) {
                    const unsigned long long at64 = lay.pool_next ? atomicAdd(lay.pool_next, (unsigned long long)kPoolChunk) : ~0ull;
                    const bool got = at64 <= lay.pool_rows && lay.pool_rows - at64 >= kPoolChunk;
                    L.first_flag = got ? (uint32_t)at64 : 0xFFFFFFFFu;
                    if (got) L.pool_chunk = (uint32_t)at64 + 1;
                }
                WG_SYNC();
                const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.first_flag);
                WG_SYNC();
                if (at == 0xFFFFFFFFu) {
                    bail = true;
                    break;
                }
                chunk = at + 1;
                row = 0;
                row_budget = kPoolChunk;
                set_rows = 0;   // (the open set stays behind in the rows the stream has left)
                col0 = 0;
            }
            uint32_t *const rows_at = (chunk ? lay.tok + (lay.pool_first + (chunk - 1)) * kT : tok) + row * kT + col0;
            uint32_t nlanes = 0, sub_next = sub_bits, round_bytes = 0;
            const uint32_t used = decode_round_tokens(L, c, win_bit0, sub_bits, rows_at, min((uint32_t)kDT, (uint32_t)kT - col0),
                                                      &nlanes, &sub_next, &round_bytes);
            // A sub-sequence near the round's front that ran into the token or byte limit ends the round there: the
            // round owns a full set of rows for a few sub-sequences' worth of input.  Where the stream has rows to
            // spare that is the cheaper way; where it has not (a short stream, the end of a share) the round is not
            // recorded: the position goes back and the same bits are decoded again with sub-sequences half as long.
            const bool front_full = (nlanes >> 31) && (nlanes & 0x7FFFFFFFu) * 4 < (uint32_t)kDT;
            nlanes &= 0x7FFFFFFFu;
            if (front_full && sub_bits > kDecSubMin && row + 2 * (kHdrRows + kTokCap) > row_budget) {
                sub_bits = max(kDecSubMin, (((sub_bits >> 6) - 1) | 1u) << 5);
                if (tid == 0) L.bitpos = (uint64_t)(win_bit0 + (int64_t)b0);   // (a round that ends this way changed nothing else)
                WG_SYNC();
                continue;
            }
            produced += round_bytes;
            sub_bits = sub_next;
            if (tid == 0) rounds[nrounds] = RoundRec{ROUND_TOKENS, (uint32_t)row, nlanes | (col0 << 16), chunk};
            nrounds++;
            // the next round joins this one's rows when about as many columns are left as this one took
            set_rows = max(set_rows, used);
            const uint32_t left = (uint32_t)kT - (col0 + nlanes);
            if (left >= 8 && left >= min(128u, nlanes + nlanes / 4)) {
                col0 += nlanes;
            } else {
                row += set_rows;
                set_rows = 0;
                col0 = 0;
            }
        }
        const int st = L.status;
        const int32_t stored = L.stored_len;
        // (read in front of the barrier: behind it the first thread moves L.bitpos on, and a wave that saw the
        // new value would find the block longer than what is left of the body and leave the loop alone)
        const uint64_t src_byte = L.bitpos >> 3;
        WG_SYNC();
        if (st < 0 && stored >= 0) {
            // stored block body (src/decompress/mod.rs:305-346): the expander copies it
            if ((uint64_t)stored > c.body_len - src_byte) {
                bail = true;   // ShortInput
                break;
            }
            produced += (uint64_t)stored;
            if (tid == 0) {
                rounds[nrounds] = RoundRec{ROUND_STORED, (uint32_t)stored, (uint32_t)src_byte, (uint32_t)(src_byte >> 32)};
                L.stored_len = -1;
                L.bitpos += (uint64_t)stored * 8;
                if (L.final_block) L.status = RSX_SUCCESS;
            }
            nrounds++;
            WG_SYNC();
            if (L.status >= 0) break;
            continue;
        }
        if (st >= 0) break;
    }
    if (tid == 0) {
        // A part that has another part of its stream behind it must end between two blocks with more blocks to
        // come: a final block that happens to end exactly at the joint ends the STREAM there (the bytes behind it
        // are not part of it, src/decompress/mod.rs:246-249), which only the one-kernel decoder can report.
        const bool ok = !bail && L.status == RSX_SUCCESS && !(open_end && L.final_block);
        if (ok) rounds[nrounds] = RoundRec{ROUND_END, 0, 0, 0};
        status[item] = ok ? kStatusPending : kStatusFallback;
        body_end[item] = ci.hdr + ((L.bitpos + 7) >> 3);
        if (parts.bytes) parts.bytes[item] = produced;
    }
    SPLIT_T_END();
}

// LDS of the second kernel: the output ring, the copy phase's pointer table (named win: copy_round is
// shared with inflate_kernel) and the match queue.
#ifndef RSX_XRING
#define RSX_XRING 16384
#endif
constexpr uint32_t kXRing = RSX_XRING, kXRingMask = kXRing - 1;   // the second kernel's ring
static_assert((kXRing & kXRingMask) == 0 && kXRing >= 4096 && kXRing <= 65536, "a power of two that holds a group");
#ifndef RSX_XMATCHES
#define RSX_XMATCHES 1792
#endif
constexpr uint32_t kXMaxMatches = RSX_XMATCHES;   // match queue entries per group
static_assert(kTokCap <= kXMaxMatches, "one sub-sequence must fit the match queue");
struct XMatchQueue {
    uint16_t dst[kXMaxMatches];   // as MatchQueue
};
struct __align__(16) ExpandLds {
    uint32_t win[kWinWords + 4];
    uint8_t ring[kXRing];
    XMatchQueue mq;
    uint32_t scan_carry[3 * (kT / 64) + 4];
    uint32_t stat_copy;
    unsigned long long stat_t[4];
    uint32_t num_matches;
    uint32_t round_start_lo, round_end_lo;
    uint32_t group_end, base_bytes, base_nm;
    uint32_t bail;
    uint32_t lane_at[kT];    // per sub-sequence of the round: output bytes in front of it (from the round's start)
    uint16_t lane_mi[kT];    // matches in front of it
    uint8_t lane_steps[kT];  // its tokens
    SPLIT_T_FIELD
};

#ifndef RSX_XOCC
#define RSX_XOCC 5
#endif
#ifndef RSX_XGROUP
#define RSX_XGROUP (16384 - 16)
#endif
constexpr uint32_t kXGroupOut = RSX_XGROUP;   // output bytes per group of the second kernel
static_assert(kLaneBytesStop + 258 <= kXGroupOut && kXGroupOut + 16 <= kXRing, "one sub-sequence fits a group, a group the ring");
// A stream that was cut into parts whose matches reach across the joints (zlib's sync flush keeps the
// window) cannot be expanded part beside part, but its tokens are all there: a second launch in which one
// workgroup takes a whole stream and walks the token rows of its parts one after the other, the ring and
// the output position running through.
struct ExpandChain {
    const uint32_t *parent_first;   // null: every item on its own; else the first part of every stream ...
    const uint32_t *part_parent;    // ... and the stream of every part
    const uint64_t *stream_out_off; // the streams' output slots
    int32_t *stream_status;         // kStatusChain -> RSX_SUCCESS / kStatusFallback
    uint64_t *stream_out_sizes;
    uint32_t num_streams;
    const uint8_t *inner;           // (first launch) per part: it has a part of its own stream in front of it
};
template <bool kChained>   // (two kernels by name: the profiler's statistics keep their launches apart)
__global__ __launch_bounds__(kT, RSX_XOCC) void inflate_expand_kernel(
    int format, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off, uint64_t *__restrict__ out_sizes,
    int32_t *__restrict__ status, SplitLayout lay, const uint64_t *__restrict__ in_hi,
    const uint64_t *__restrict__ out_hi, const uint64_t *__restrict__ in_base_p, const uint32_t *__restrict__ count,
    ExpandChain chain)
{
    __shared__ ExpandLds L;
    const uint32_t item = blockIdx.x;
    constexpr bool chained = kChained;
    uint32_t seg, seg_end;   // the items whose token rows this workgroup walks
    uint64_t ob, oe;
    if (chained) {
        if (item >= chain.num_streams || chain.stream_status[item] != kStatusChain) return;
        const uint32_t np = *count;
        seg = chain.parent_first[item];
        for (seg_end = seg; seg_end < np && chain.part_parent[seg_end] == item;) seg_end++;
        ob = chain.stream_out_off[item];
        oe = chain.stream_out_off[item + 1];
    } else {
        if (item >= n || (count && item >= *count)) return;
        if (status[item] != kStatusPending) return;   // (uniform)
        seg = item;
        seg_end = item + 1;
        ob = out_off[item];
        oe = out_hi[item];
    }
    const int tid = threadIdx.x;
    StreamCtx c;
    c.in_alloc = in;
    c.in_alloc_len = 0;
    c.out = out + ob;
    c.out_cap = oe - ob;
    c.shift = (uint32_t)((uintptr_t)c.out & 15);
    const uint64_t in_base = *in_base_p;
    const uint32_t cap = (uint32_t)c.out_cap;

    if (tid == 0) {
        L.bail = 0;
        L.stat_copy = 0;
        L.round_start_lo = 0;
        L.round_end_lo = 0;
        L.num_matches = 0;
    }
    uint32_t flushed = 0;   // output positions below this are in global memory (uniform)
    uint32_t pos = 0;       // output position reached (uniform)
    bool bail = false, reaches_back = false;
    SPLIT_T_BEGIN();
    WG_SYNC();
    auto flush_upto = [&](uint32_t upto_pos) {
        uint32_t upto = ((upto_pos + c.shift) & ~15u);
        upto = upto > c.shift ? upto - c.shift : 0;
        if (upto < flushed) upto = flushed;
        flush_range(L, c, flushed, upto);
        flushed = upto;
    };
    for (; seg < seg_end && !bail; seg++) {
    const uint64_t ib = in_off[seg], ie = in_hi[seg];
    // (the first kernel accepted the container: its header is parsed again for the body's place)
    const ContainerInfo ci = parse_container(format, in + ib, ie - ib);
    c.body = in + ib + ci.hdr;
    c.body_len = (ie - ib) - ci.hdr - ci.trailer;
    const uint32_t *const tok = lay.tok + lay.first_row(ib - in_base, seg) * kT;
    const RoundRec *const rounds = lay.rounds + lay.first_round(ib - in_base, seg);
    for (uint64_t r = 0; !bail; r++) {
        unsigned long long tb = SPLIT_T_NOW();
        (void)tb;   // (read only by the phase timers of development builds)
        const RoundRec rec = rounds[r];
        if (rec.kind == ROUND_END) break;
        if (rec.kind == ROUND_STORED) {
            const uint32_t len = rec.a;
            const uint64_t src_byte = rec.b | ((uint64_t)rec.c << 32);
            if (len > cap - pos) {
                bail = true;   // InsufficientSpace
                break;
            }
            flush_upto(pos);
            const uint8_t *sp = c.body + src_byte;
            for (uint32_t done = 0; done < len;) {
                const uint32_t piece = len - done < kXGroupOut ? len - done : kXGroupOut;
                for (uint32_t k = tid; k < piece; k += kT) L.ring[(pos + k + c.shift) & kXRingMask] = sp[done + k];
                WG_SYNC();
                pos += piece;
                done += piece;
                flush_upto(pos);
                WG_SYNC();
            }
            continue;
        }
        // a round of tokens: its sub-sequences go through the ring in groups that fit it
        const uint32_t nlanes = rec.b & 0xFFFF;   // (its columns start at rec.b >> 16: rounds share rows)
        const uint32_t *const rows = (rec.c ? lay.tok + (lay.pool_first + (rec.c - 1) + rec.a) * kT : tok + (uint64_t)rec.a * kT) + (rec.b >> 16);
        const uint32_t info = (uint32_t)tid < nlanes ? rows[tid] : 0u;
        const uint32_t ntok = info & 0xFF, nm = (info >> 8) & 0xFF, bytes = info >> 16;
        uint32_t exb, exn;
        wg_exclusive_scan2(L, bytes, nm, &exb, &exn);
        L.lane_at[tid] = exb;
        L.lane_mi[tid] = (uint16_t)exn;
        L.lane_steps[tid] = (uint8_t)ntok;
        if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(8, t - tb); tb = t; }
        uint32_t g0 = 0;
        while (g0 < nlanes) {
            if (RSX_SPLIT_TIMERS && tid == 0) { tb = SPLIT_T_NOW(); SPLIT_T_ADD(12, 1); }
            if ((uint32_t)tid == g0) {
                L.base_bytes = exb;
                L.base_nm = exn;
                L.group_end = nlanes;
            }
            WG_SYNC();
            const uint32_t bb = L.base_bytes, bn = L.base_nm;
            const bool fits = exb + bytes - bb <= kXGroupOut && exn + nm - bn <= kXMaxMatches;
            if ((uint32_t)tid > g0 && (uint32_t)tid < nlanes && !fits) atomicMin(&L.group_end, (uint32_t)tid);
            WG_SYNC();
            const uint32_t g1 = L.group_end;
            const uint32_t round_start = pos;
            if ((uint32_t)tid >= g0 && (uint32_t)tid < g1) {
                // InsufficientSpace somewhere in this sub-sequence?
                const uint32_t op0 = round_start + (exb - bb);
                if (op0 > cap || bytes > cap - op0) L.bail = 1;
                if ((uint32_t)tid == g1 - 1) {
                    L.round_start_lo = round_start;
                    L.round_end_lo = round_start + (exb - bb) + bytes;
                    L.num_matches = exn - bn + nm;
                }
            }
            WG_SYNC();
            // every lane takes pieces of kSegSteps steps: piece p of sub-sequence l starts with the totals of
            // header row p; neighbouring lanes take the same piece of neighbouring sub-sequences (one row, one
            // contiguous load)
            if (!L.bail) {
                const uint32_t G = g1 - g0;
                bool err = false;
                for (uint32_t u = tid; u < G * (kTokCap / kSegSteps); u += kT) {
                    const uint32_t l = g0 + u % G, p = u / G;
                    const uint32_t steps_l = L.lane_steps[l];
                    if (p * kSegSteps >= steps_l) continue;
                    const uint32_t nslot = min(kSegSteps, steps_l - p * kSegSteps);
                    const uint32_t *col = rows + (kHdrRows + p * kSegSteps) * kT + l;
                    uint32_t t16[kSegSteps];
#pragma unroll
                    for (uint32_t j = 0; j < kSegSteps; j++) t16[j] = j < nslot ? col[j * kT] : 0u;
                    const uint32_t before = p ? rows[p * kT + l] : 0u;
                    uint32_t op = round_start + (L.lane_at[l] - bb) + (before & 0xFFFF);
                    uint32_t mi = L.lane_mi[l] - bn + (before >> 16);
                    // A literal and a match both leave the token's low byte where they start (the literal itself,
                    // or the low half of distance - 1), so the common part of a step has no select; what only a
                    // match does stands behind one test.
#pragma unroll
                    for (uint32_t j = 0; j < kSegSteps; j++) {
                        const uint32_t t = t16[j];
                        const bool act = j < nslot, is_m = (int32_t)t < 0;   // (slots behind the last token hold 0)
                        const uint32_t at = op + c.shift;
                        if (act) L.ring[at & kXRingMask] = (uint8_t)t;
                        if (is_m) {
                            const uint32_t d1 = t & 0x7FFF;
                            err = err || d1 >= op;   // the distance reaches in front of the output
                            L.mq.dst[mi] = (uint16_t)(op - round_start);
                            L.ring[(at + 1) & kXRingMask] = (uint8_t)(d1 >> 8);
                            L.ring[(at + 2) & kXRingMask] = (uint8_t)(t >> 15);
                        }
                        op += is_m ? ((t >> 15) & 0xFF) + 3 : act ? 1u : 0u;
                        mi += is_m ? 1u : 0u;
                    }
                }
                if (err) L.bail = 2;   // (nothing else sets 2; whoever sets 1 beside it loses nothing: both end the item)
            }
            // (the bytes that the group before flushed to the output are read back by this group's far
            // matches, by other waves than the ones that stored them: every wave's stores have arrived
            // before anyone passes this barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WG_SYNC();
            if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(9, t - tb); tb = t; }
            if (L.bail) {
                reaches_back = L.bail == 2;
                bail = true;
                break;
            }
            copy_round(L, c);
            if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(10, t - tb); tb = t; }
            pos = L.round_end_lo;
            flush_upto(pos);
            WG_SYNC();
            if (RSX_SPLIT_TIMERS && tid == 0) { unsigned long long t = SPLIT_T_NOW(); SPLIT_T_ADD(11, t - tb); tb = t; }
            g0 = g1;
        }
    }
    }
    if (!bail) flush_range(L, c, flushed, pos);
    if (tid == 0) {
        if (chained) {
            chain.stream_status[item] = bail ? kStatusFallback : RSX_SUCCESS;
            chain.stream_out_sizes[item] = bail ? 0 : pos;
        } else {
            // a distance that reaches in front of a part which has a part in front of it may be a good one
            const bool again = bail && reaches_back && chain.inner && chain.inner[item];
            status[item] = !bail ? RSX_SUCCESS : again ? kStatusChain : kStatusFallback;
            out_sizes[item] = bail ? 0 : pos;
        }
    }
    SPLIT_T_END();
}

// zlib / gzip trailer check (src/decompress/mod.rs:1104-1120, :1202-1233): the
// checksum of the produced bytes must match the stored one (and ISIZE for gzip).
__global__ void verify_trailer_kernel(int format, const uint8_t *__restrict__ in,
                                      const uint64_t *__restrict__ in_off, uint32_t n,
                                      const uint64_t *__restrict__ body_end,
                                      const uint32_t *__restrict__ sums,
                                      uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status,
                                      uint64_t *__restrict__ in_used)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // bytes of the item the member took, header and trailer included (0 for a failed item)
    if (in_used) in_used[i] = status[i] == RSX_SUCCESS ? body_end[i] + (format == RSX_FORMAT_ZLIB ? 4 : 8) : 0;
    if (status[i] != RSX_SUCCESS) return;
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
        if (in_used) in_used[i] = 0;
    }
}

// =====================================================================================
// Streams cut into parts.  One stream is decoded by one workgroup, so a batch of a few large streams
// leaves most of the GPU idle.  The reference's compressor cuts inputs above 256 KiB into chunks that
// it compresses independently and joins with sync-flush markers (src/compress/mod.rs:699-772: an empty
// stored block, byte aligned, 00 00 FF FF), and so does this engine's; a decoder that stands between two
// blocks at a byte boundary can start there without knowing anything of what came before, as long as no
// match reaches back across the joint.  For batches of few, large streams the input is therefore
// searched for the marker (the first one in every 32 KiB tile), and every stretch between two finds is
// decoded as a part of its own by the two kernels above.  The marker is only a hint: a part is accepted
// when its decoder arrives, between two blocks, at exactly the byte where the next part starts (the last
// part: at the final block's end), and when no match of it reaches in front of its first byte.  Then the
// parts' sizes give their places in the stream's output slot, the second kernel expands them side by
// side, and the stream is reported like any other.  A stream with a part that fails any of this is
// decoded again as a whole by the one-kernel decoder, which also reports every error.
constexpr uint32_t kPartTile = 32768;        // at most one cut per tile of the flat input
constexpr int32_t kStatusSkip = -4;          // a part whose stream is left to inflate_kernel
constexpr uint64_t kNoBody = ~0ull;

// per stream: the DEFLATE body [body_lo, body_hi) as offsets into `in` (kNoBody: the container is damaged)
__global__ void part_items_kernel(int format, const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off,
                                  uint32_t n, uint64_t *__restrict__ body_lo, uint64_t *__restrict__ body_hi)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t ib = in_off[i], ie = in_off[i + 1];
    const ContainerInfo ci = parse_container(format, in + ib, ie - ib);
    const bool ok = ci.status < 0 && (uint64_t)ci.hdr + ci.trailer <= ie - ib;
    body_lo[i] = ok ? ib + ci.hdr : kNoBody;
    body_hi[i] = ok ? ie - ci.trailer : kNoBody;
}

// one workgroup per tile: the first marker that starts in the tile, the stream it lies in, and whether the
// byte behind it lies inside that stream's body (then a part can start there)
__global__ __launch_bounds__(kT) void part_find_kernel(
    const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n, uint64_t total_in,
    const uint64_t *__restrict__ body_lo, const uint64_t *__restrict__ body_hi, uint64_t *__restrict__ tile_first,
    uint32_t *__restrict__ tile_item, uint32_t *__restrict__ tile_flag)
{
    __shared__ uint32_t first;   // offset of the earliest marker from the tile's first word
    const uint32_t b = blockIdx.x;
    const uint64_t base = in_off[0];
    const uint64_t t0 = base + (uint64_t)b * kPartTile;   // the tile is in[t0, t1)
    const uint64_t end = base + total_in;
    const uint64_t t1 = t0 + kPartTile < end ? t0 + kPartTile : end;
    if (threadIdx.x == 0) first = ~0u;
    WG_SYNC();
    // aligned 16-byte units from the one that holds in[t0] on (a unit that holds a byte of the input is
    // readable: it does not cross a page), one unit per lane and turn, plus the word behind it
    const uintptr_t a0 = (uintptr_t)(in + t0) & ~(uintptr_t)15;
    const uintptr_t a_end = (uintptr_t)(in + end);
    const uint32_t lead = (uint32_t)((uintptr_t)(in + t0) - a0);   // bytes of the first unit in front of the tile
    const uint32_t nunits = (uint32_t)((t1 - t0 + lead + 15) / 16);
    for (uint32_t k = threadIdx.x; k < nunits; k += kT) {
        const uintptr_t ua = a0 + 16ull * k;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 q = *(const __attribute__((address_space(1))) u32x4 *)ua;
        const uint32_t next = ua + 16 < a_end ? *(const __attribute__((address_space(1))) uint32_t *)(ua + 16) : 0u;
        const uint32_t w[5] = {q.x, q.y, q.z, q.w, next};
#pragma unroll
        for (uint32_t d = 0; d < 4; d++) {
            const uint64_t v = ((uint64_t)w[d + 1] << 32) | w[d];
#pragma unroll
            for (uint32_t sft = 0; sft < 4; sft++) {
                const uint32_t at = 16 * k + 4 * d + sft;   // from a0
                // 00 00 FF FF starting at this byte, inside the tile and whole inside the input
                if ((uint32_t)(v >> (8 * sft)) == 0xFFFF0000u && at >= lead && t0 + (at - lead) < t1 &&
                    t0 + (at - lead) + 4 <= end)
                    atomicMin(&first, at - lead);
            }
        }
    }
    WG_SYNC();
    if (threadIdx.x == 0) {
        const uint32_t f = first;
        uint64_t start = 0;
        uint32_t item = 0, flag = 0;
        if (f != ~0u) {
            const uint64_t at = t0 + f;   // where the marker starts
            start = at + 4;
            // the stream that holds the marker's first byte: the last i with in_off[i] <= at
            uint32_t lo_i = 0, hi_i = n;   // in_off[lo_i] <= at < in_off[hi_i]
            while (hi_i - lo_i > 1) {
                const uint32_t mid = lo_i + (hi_i - lo_i) / 2;
                if (in_off[mid] <= at) lo_i = mid; else hi_i = mid;
            }
            item = lo_i;
            const uint64_t bl = body_lo[item], bh = body_hi[item];
            flag = bl != kNoBody && at >= bl && start < bh;
        }
        tile_first[b] = start;
        tile_item[b] = item;
        tile_flag[b] = flag;
    }
}

// exclusive prefix sums of the tiles' flags (one workgroup); *count = streams + cuts = number of parts
__global__ __launch_bounds__(1024) void part_scan_kernel(const uint32_t *__restrict__ tile_flag, uint32_t ntiles,
                                                         uint32_t n, uint32_t *__restrict__ tile_pre,
                                                         uint32_t *__restrict__ count)
{
    __shared__ uint32_t wsum[16];
    // every thread takes a run of consecutive tiles: one scan over the threads' sums, whatever the tile count
    const uint32_t per = (ntiles + 1023) / 1024;
    const uint32_t lo = min(threadIdx.x * per, ntiles), hi = min(lo + per, ntiles);
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += tile_flag[i];
    uint32_t x = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(x, o);
        if ((int)(threadIdx.x & 63) >= o) x += y;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
    WG_SYNC();
    uint32_t before = x - mine, carry = 0;
    for (uint32_t k = 0; k < 16; k++) {
        if (k < (threadIdx.x >> 6)) before += wsum[k];
        carry += wsum[k];
    }
    for (uint32_t i = lo; i < hi; i++) {
        tile_pre[i] = before;
        before += tile_flag[i];
    }
    if (threadIdx.x == 0) {
        tile_pre[ntiles] = carry;
        count[0] = n + carry;
        count[1] = 0;   // streams taken out behind the first kernel
        count[2] = 0;   // ... behind the second
    }
}

// The parts in the order of their first bytes: stream starts and cuts merged.  Cut b comes behind the
// starts of streams 0..item and behind the cuts of earlier tiles; the start of stream i comes behind i
// earlier starts and the cuts that lie in earlier streams.
__global__ void part_scatter_kernel(const uint64_t *__restrict__ in_off, uint32_t n, uint64_t total_in,
                                    uint32_t ntiles, const uint64_t *__restrict__ body_lo,
                                    const uint64_t *__restrict__ tile_first, const uint32_t *__restrict__ tile_item,
                                    const uint32_t *__restrict__ tile_flag, const uint32_t *__restrict__ tile_pre,
                                    uint64_t *__restrict__ part_lo, uint32_t *__restrict__ part_parent)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntiles) {
        if (!tile_flag[t]) return;
        const uint32_t idx = tile_pre[t] + tile_item[t] + 1;
        part_lo[idx] = tile_first[t];
        part_parent[idx] = tile_item[t];
        return;
    }
    const uint32_t i = t - ntiles;
    if (i >= n) return;
    const uint64_t rel = in_off[i] - in_off[0];
    uint32_t before;   // cuts that lie in streams in front of this one
    if (rel >= total_in) {
        before = tile_pre[ntiles];
    } else {
        const uint32_t tb = (uint32_t)(rel / kPartTile);
        before = tile_pre[tb] + ((tile_flag[tb] && tile_item[tb] < i) ? 1u : 0u);
    }
    const uint64_t bl = body_lo[i];
    part_lo[i + before] = bl != kNoBody ? bl : in_off[i];
    part_parent[i + before] = i;
}

// per part: where its input ends, whether it is the last of its stream; per stream: its first part
__global__ void part_finish_kernel(uint32_t n, const uint32_t *__restrict__ count, const uint64_t *__restrict__ body_hi,
                                   uint64_t *__restrict__ part_lo, const uint32_t *__restrict__ part_parent,
                                   uint64_t *__restrict__ part_hi, uint8_t *__restrict__ part_open,
                                   int32_t *__restrict__ part_status, uint32_t *__restrict__ parent_first,
                                   uint8_t *__restrict__ part_inner)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t np = *count;
    if (s >= np) return;
    const uint32_t p = part_parent[s];
    if (p >= n) {   // (cannot happen: every slot below *count is written by part_scatter_kernel)
        part_hi[s] = part_lo[s] = 0;
        part_open[s] = 0;
        part_inner[s] = 0;
        part_status[s] = kStatusSkip;
        return;
    }
    const bool last = s + 1 == np || part_parent[s + 1] != p;
    const uint64_t bh = body_hi[p];
    part_hi[s] = bh == kNoBody ? part_lo[s] : last ? bh : part_lo[s + 1];
    part_open[s] = last ? 0 : 1;
    part_status[s] = bh == kNoBody ? kStatusFallback : kStatusPending;   // (the first kernel overwrites it)
    const bool first = s == 0 || part_parent[s - 1] != p;
    part_inner[s] = first ? 0 : 1;
    if (first) parent_first[p] = s;
}

// Behind the first kernel, per stream: every part must have decoded, every part but the last must have
// used its input to the last byte, and the sum of the parts' bytes must fit the stream's slot.  Then the
// parts get their places in the slot; else all of them are taken out and the stream is marked for
// inflate_kernel.
__global__ void part_place_kernel(const uint64_t *__restrict__ out_off, uint32_t n, const uint32_t *__restrict__ count,
                                  const uint32_t *__restrict__ parent_first, const uint32_t *__restrict__ part_parent,
                                  const uint64_t *__restrict__ part_lo, const uint64_t *__restrict__ part_hi,
                                  const uint64_t *__restrict__ part_used, const uint64_t *__restrict__ part_bytes,
                                  int32_t *__restrict__ part_status, uint64_t *__restrict__ part_olo,
                                  uint64_t *__restrict__ part_ohi, int32_t *__restrict__ status,
                                  uint32_t *__restrict__ taken_out)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t np = *count, s0 = parent_first[p];
    const uint64_t cap = out_off[p + 1] - out_off[p];
    bool ok = true;
    uint64_t total = 0;
    uint32_t s1 = s0;
    for (; s1 < np && part_parent[s1] == p; s1++) {
        const bool last = s1 + 1 == np || part_parent[s1 + 1] != p;
        if (part_status[s1] != kStatusPending) ok = false;
        else if (!last && part_used[s1] != part_hi[s1] - part_lo[s1]) ok = false;
        else total += part_bytes[s1];
        if (total > cap) ok = false;
        if (!ok) break;
    }
    if (ok) {
        uint64_t at = out_off[p];
        for (uint32_t s = s0; s < s1; s++) {
            part_olo[s] = at;
            at += part_bytes[s];
            part_ohi[s] = at;
        }
        status[p] = kStatusPending;
    } else {
        for (uint32_t s = s0; s < np && part_parent[s] == p; s++) {
            part_status[s] = kStatusSkip;
            part_olo[s] = part_ohi[s] = out_off[p];
        }
        status[p] = kStatusFallback;
        atomicAdd(taken_out, 1u);
    }
}

// Behind the second kernel, per stream: all parts expanded -> the stream's size, status and the end of its
// DEFLATE body; else it is marked for inflate_kernel.
__global__ void part_reduce_kernel(const uint64_t *__restrict__ in_off, uint32_t n, const uint32_t *__restrict__ count,
                                   const uint32_t *__restrict__ parent_first, const uint32_t *__restrict__ part_parent,
                                   const uint64_t *__restrict__ part_lo, const uint64_t *__restrict__ part_used,
                                   const int32_t *__restrict__ part_status, const uint64_t *__restrict__ part_sizes,
                                   uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status,
                                   uint64_t *__restrict__ body_end, uint32_t *__restrict__ taken_out)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    out_sizes[p] = 0;
    body_end[p] = 0;
    if (status[p] != kStatusPending) return;   // (already marked for inflate_kernel)
    const uint32_t np = *count;
    bool ok = true, reaches_back = false;
    uint64_t total = 0;
    uint32_t last = parent_first[p];
    for (uint32_t s = last; s < np && part_parent[s] == p; s++) {
        if (part_status[s] == kStatusChain) reaches_back = true;
        else if (part_status[s] != RSX_SUCCESS) ok = false;
        total += part_sizes[s];
        last = s;
    }
    if (ok && !reaches_back) {
        status[p] = RSX_SUCCESS;
        out_sizes[p] = total;
        body_end[p] = (part_lo[last] - in_off[p]) + part_used[last];
    } else if (ok) {
        // parts that copy from the parts in front of them: the stream is expanded again, by one workgroup
        status[p] = kStatusChain;
        body_end[p] = (part_lo[last] - in_off[p]) + part_used[last];
        atomicAdd(taken_out, 1u);
    } else {
        status[p] = kStatusFallback;
        atomicAdd(taken_out, 1u);
    }
}

// Scratch of rsx_batch_decompress_device_noalloc: every GPU the library runs on holds its own copy
// of this array until the process ends; host code reaches it only through the address that
// inflate_static_scratch() looks up.
constexpr size_t kStaticScratchBytes = 1u << 20;
static __device__ __align__(256) uint8_t scratch[kStaticScratchBytes];

}  // namespace

int inflate_static_scratch(rsx_engine *e)
{
    void *ptr = nullptr;
    RSX_HIP(e, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    e->static_scratch = static_cast<uint8_t *>(ptr);
    e->static_scratch_size = kStaticScratchBytes;
    return RSX_OK;
}

// One item decoded by the one-kernel decoder in its resumable mode (see ResumeArgs).  ctl = ten 64-bit words of
// device memory: in_off[2], out_off[2] (the slot starts with the history), hist_len, end_bits, out_size,
// body_end, start_bit (low half), status (low half); d_in / d_out are the bases the offsets refer to.
int inflate_resume_launch(rsx_engine *e, const uint8_t *d_in, uint64_t in_len, uint8_t *d_out, uint64_t *ctl)
{
    const ResumeArgs ra{reinterpret_cast<const uint32_t *>(ctl + 8), ctl + 4, ctl + 5};
    inflate_kernel<false><<<1, kT, 0, e->stream>>>(RSX_FORMAT_DEFLATE, d_in, ctl, 1u, in_len, d_out, ctl + 2, ctl + 6,
                                                    reinterpret_cast<int32_t *>(ctl + 9), ctl + 7, nullptr, 0, ctl + 1,
                                                    ctl + 3, ctl, nullptr, ra);
    RSX_HIP(e, hipGetLastError());
    return RSX_OK;
}

static size_t inflate_common_ws_bytes(size_t n, uint64_t total_out)
{
    return align_up(n * sizeof(uint64_t), 256) + align_up(n * sizeof(uint32_t), 256) +
           checksum_ws_bytes(n, total_out) + 256 + 1024;   // (256: the counter of streams decoded twice)
}

// Number of tiles the input is searched in for cuts, 0 when the batch is decoded stream by stream: cutting
// pays for few, large streams (many streams fill the GPU by themselves, and every part costs token rows).
#ifndef RSX_PARTS_MAX_N
#define RSX_PARTS_MAX_N 65536
#endif
static uint32_t inflate_part_tiles(size_t n, uint64_t total_in)
{
    if (n > RSX_PARTS_MAX_N || total_in < (1u << 20) || total_in > (1ull << 32) || total_in / n < (256u << 10)) return 0;
    return (uint32_t)((total_in + kPartTile - 1) / kPartTile);
}

static size_t inflate_parts_ws_bytes(size_t n, uint32_t ntiles)
{
    const size_t cap = n + ntiles;
    return 3 * align_up(n * 8, 256) + align_up(ntiles * 8, 256) + 3 * align_up((ntiles + 1) * 4, 256) + 256 +
           7 * align_up(cap * 8, 256) + 2 * align_up(cap * 4, 256) + 2 * align_up(cap, 256);
}

size_t inflate_ws_bytes(size_t n, uint64_t total_in, uint64_t total_out, bool split)
{
    size_t b = inflate_common_ws_bytes(n, total_out);
    if (split) {
        const uint32_t ntiles = inflate_part_tiles(n, total_in);
        const size_t items = n + ntiles;
        b += align_up((split_tok_rows(items, total_in) + split_pool_rows(total_in)) * kT * sizeof(uint32_t), 256) +
             align_up(split_round_recs(items, total_in) * sizeof(RoundRec), 256);
        if (ntiles) b += inflate_parts_ws_bytes(n, ntiles);
    }
    return b;
}

int inflate_launch(rsx_engine *e, int format, const uint8_t *in, const uint64_t *in_off, size_t n,
                   uint64_t total_in, uint64_t total_out, uint64_t max_out, uint8_t *out,
                   const uint64_t *out_off, uint64_t *out_sizes, int32_t *status, bool split,
                   const InflateSlices *slices)
{
    if (n == 0) return RSX_OK;
    const int fmt = format & kFormatMask;   // (`format` itself goes to the kernels that judge the container header)
    const uint64_t *in_hi = slices ? slices->in_hi : in_off + 1;
    const uint64_t *out_hi = slices ? slices->out_hi : out_off + 1;
    const uint64_t *in_base_p = slices ? slices->in_base : in_off;
    uint64_t *in_used = slices ? slices->in_used : nullptr;
    // (output positions are 32-bit in the kernels, and a round adds up to 16 KiB before it is checked)
    if (n > 0x7FFFFFFFull || max_out >= (1ull << 32) - (1ull << 16)) {
        e->err = "batch too large (output slots must be < 4 GiB - 64 KiB, n < 2^31)";
        return RSX_E_INVALID;
    }
    uint64_t *body_end = static_cast<uint64_t *>(ws_take(e, n * sizeof(uint64_t)));
    uint32_t *sums = static_cast<uint32_t *>(ws_take(e, n * sizeof(uint32_t)));
    // streams cut into parts: only for the usual layout (an item's slices could overlap a neighbour's)
    const uint32_t ntiles = split && !slices && e->inflate_parts ? inflate_part_tiles(n, total_in) : 0u;
    const size_t items = n + (split && !slices ? inflate_part_tiles(n, total_in) : 0u);   // (as inflate_ws_bytes counts)
    SplitLayout lay{nullptr, nullptr, 0, 0, nullptr};
    if (split) {
        lay.pool_first = split_tok_rows(items, total_in);
        lay.pool_rows = (uint32_t)split_pool_rows(total_in);
        lay.tok = static_cast<uint32_t *>(ws_take(e, (lay.pool_first + lay.pool_rows) * kT * sizeof(uint32_t)));
        lay.rounds = static_cast<RoundRec *>(ws_take(e, split_round_recs(items, total_in) * sizeof(RoundRec)));
    }
    if (!body_end || !sums || (split && (!lay.tok || !lay.rounds))) {
        e->err = "workspace exhausted";
        return RSX_E_NOMEM;
    }
    const uint32_t grid = (uint32_t)n;
    // the per-stream timers of RSX_INFLATE_STATS allocate and synchronise: never on the static scratch, whose
    // callers must not allocate
    const bool stats = !split && e->inflate_stats && e->arena != e->static_scratch;
    // how many streams the one-kernel decoder takes behind the two (rsx_engine_last_fallbacks)
    uint32_t *taken = split ? static_cast<uint32_t *>(ws_take(e, 256)) : nullptr;
    e->d_last_fallbacks = nullptr;
    if (split) {
        if (!taken) {
            e->err = "workspace exhausted";
            return RSX_E_NOMEM;
        }
        RSX_HIP(e, hipMemsetAsync(taken, 0, 16, e->stream));
        e->d_last_fallbacks = taken;
        lay.pool_next = reinterpret_cast<unsigned long long *>(taken + 2);
    }
    (void)hipEventRecord(e->ev_begin, e->stream);
    e->d_last_parts = nullptr;
    if (ntiles) {
        const size_t cap = n + ntiles;
        auto take64 = [&](size_t k) { return static_cast<uint64_t *>(ws_take(e, k * 8)); };
        auto take32 = [&](size_t k) { return static_cast<uint32_t *>(ws_take(e, k * 4)); };
        uint64_t *body_lo = take64(n), *body_hi = take64(n), *tile_first = take64(ntiles);
        uint32_t *parent_first = take32(2 * n);   // (8 bytes per stream reserved)
        uint32_t *tile_item = take32(ntiles + 1), *tile_flag = take32(ntiles + 1), *tile_pre = take32(ntiles + 1);
        uint32_t *count = take32(64);
        uint64_t *part_lo = take64(cap), *part_hi = take64(cap), *part_olo = take64(cap), *part_ohi = take64(cap);
        uint64_t *part_used = take64(cap), *part_bytes = take64(cap), *part_sizes = take64(cap);
        uint32_t *part_parent = take32(cap);
        int32_t *part_status = reinterpret_cast<int32_t *>(take32(cap));
        uint8_t *part_open = static_cast<uint8_t *>(ws_take(e, cap));
        uint8_t *part_inner = static_cast<uint8_t *>(ws_take(e, cap));
        if (!part_inner ||!body_lo || !body_hi || !tile_first || !parent_first || !tile_item || !tile_flag || !tile_pre || !count ||
            !part_lo || !part_hi || !part_olo || !part_ohi || !part_used || !part_bytes || !part_sizes || !part_parent ||
            !part_status || !part_open) {
            e->err = "workspace exhausted";
            return RSX_E_NOMEM;
        }
        e->d_last_parts = count;
        const uint32_t pgrid = (uint32_t)cap, tb = 256;
        part_items_kernel<<<(grid + tb - 1) / tb, tb, 0, e->stream>>>(format, in, in_off, grid, body_lo, body_hi);
        part_find_kernel<<<ntiles, kT, 0, e->stream>>>(in, in_off, grid, total_in, body_lo, body_hi, tile_first,
                                                       tile_item, tile_flag);
        part_scan_kernel<<<1, 1024, 0, e->stream>>>(tile_flag, ntiles, grid, tile_pre, count);
        part_scatter_kernel<<<(pgrid + tb - 1) / tb, tb, 0, e->stream>>>(in_off, grid, total_in, ntiles, body_lo,
                                                                         tile_first, tile_item, tile_flag, tile_pre,
                                                                         part_lo, part_parent);
        part_finish_kernel<<<(pgrid + tb - 1) / tb, tb, 0, e->stream>>>(grid, count, body_hi, part_lo, part_parent, part_hi,
                                                                        part_open, part_status, parent_first, part_inner);
        // the parts are raw DEFLATE items given as slices; their token rows are laid out by their own numbers
        inflate_decode_kernel<<<pgrid, kDT, 0, e->stream>>>(RSX_FORMAT_DEFLATE, in, part_lo, pgrid, total_in,
                                                           part_status, part_used, lay, part_hi, in_off,
                                                           DecodeParts{count, part_open, part_bytes});
        part_place_kernel<<<(grid + tb - 1) / tb, tb, 0, e->stream>>>(out_off, grid, count, parent_first, part_parent,
                                                                      part_lo, part_hi, part_used, part_bytes,
                                                                      part_status, part_olo, part_ohi, status, count + 1);
        inflate_expand_kernel<false><<<pgrid, kT, 0, e->stream>>>(RSX_FORMAT_DEFLATE, in, part_lo, pgrid, out, part_olo,
                                                                  part_sizes, part_status, lay, part_hi, part_ohi, in_off,
                                                                  count, ExpandChain{nullptr, nullptr, nullptr, nullptr, nullptr, 0, part_inner});
        part_reduce_kernel<<<(grid + tb - 1) / tb, tb, 0, e->stream>>>(in_off, grid, count, parent_first, part_parent,
                                                                       part_lo, part_used, part_status, part_sizes,
                                                                       out_sizes, status, body_end, count + 2);
        // streams whose parts copy across the joints: one workgroup per stream walks the parts' token rows
        inflate_expand_kernel<true><<<grid, kT, 0, e->stream>>>(RSX_FORMAT_DEFLATE, in, part_lo, pgrid, out, part_olo,
                                                                part_sizes, part_status, lay, part_hi, part_ohi, in_off,
                                                                count, ExpandChain{parent_first, part_parent, out_off, status, out_sizes, grid, nullptr});
        inflate_kernel<false><<<grid, kT, 0, e->stream>>>(format, in, in_off, grid, total_in, out, out_off, out_sizes,
                                                          status, body_end, nullptr, kStatusFallback, in_hi, out_hi,
                                                          in_base_p, taken, ResumeArgs{});
    } else if (split) {
        inflate_decode_kernel<<<grid, kDT, 0, e->stream>>>(format, in, in_off, grid, total_in, status, body_end, lay,
                                                          in_hi, in_base_p, DecodeParts{nullptr, nullptr, nullptr});
        inflate_expand_kernel<false><<<grid, kT, 0, e->stream>>>(fmt, in, in_off, grid, out, out_off, out_sizes, status,
                                                                 lay, in_hi, out_hi, in_base_p, nullptr, ExpandChain{});
        // whatever the two left (damaged streams, slots that are too small, streams that need more
        // token space than the workspace has) is decoded again by the one-kernel decoder
        inflate_kernel<false><<<grid, kT, 0, e->stream>>>(format, in, in_off, grid, total_in, out, out_off, out_sizes,
                                                          status, body_end, nullptr, kStatusFallback, in_hi, out_hi,
                                                          in_base_p, taken, ResumeArgs{});
    } else if (stats) {
        unsigned long long *d_stats = nullptr;   // (development aid: allocates and synchronises)
        RSX_HIP(e, hipMalloc(&d_stats, n * 12 * sizeof(unsigned long long)));
        inflate_kernel<true><<<grid, kT, 0, e->stream>>>(format, in, in_off, grid, total_in, out, out_off, out_sizes,
                                                         status, body_end, d_stats, 0, in_hi, out_hi, in_base_p, nullptr, ResumeArgs{});
        (void)hipEventRecord(e->ev_end, e->stream);
        std::vector<unsigned long long> h(n * 12);
        RSX_HIP(e, hipMemcpy(h.data(), d_stats, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_stats);
        unsigned long long sum[12] = {0};
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 12; k++) sum[k] += h[i * 12 + k];
        fprintf(stderr,
                "[rsx inflate stats] streams %zu rounds %.1f spec_passes/round %.2f copy_passes/round %.2f "
                "cycles/stream: stage %.0f header %.0f decode %.0f (pass1 %.0f later %.0f prefix %.0f emit %.0f) "
                "copy %.0f flush %.0f\n",
                n, (double)sum[0] / n, (double)sum[1] / (double)sum[0], (double)sum[2] / (double)sum[0],
                (double)sum[10] / n, (double)sum[4] / n, (double)sum[5] / n, (double)sum[3] / n,
                (double)sum[7] / n, (double)sum[8] / n, (double)sum[9] / n, (double)sum[6] / n,
                (double)sum[11] / n);
    } else {
        inflate_kernel<false><<<grid, kT, 0, e->stream>>>(format, in, in_off, grid, total_in, out, out_off, out_sizes,
                                                          status, body_end, nullptr, 0, in_hi, out_hi, in_base_p, nullptr, ResumeArgs{});
    }
#if RSX_SPLIT_TIMERS
    if (split) {
        unsigned long long t[24], z[24] = {0};
        RSX_HIP(e, hipStreamSynchronize(e->stream));
        RSX_HIP(e, hipMemcpyFromSymbol(t, HIP_SYMBOL(g_split_t), sizeof t));
        RSX_HIP(e, hipMemcpyToSymbol(HIP_SYMBOL(g_split_t), z, sizeof z));
        const double d = (double)n;
        fprintf(stderr, "[rsx split] per stream: decode rounds %.1f passes %.1f cycles: stage %.0f header %.0f warm %.0f scan %.0f "
                        "passes %.0f tail %.0f (%.1f headers: read %.0f, tables %.0f) | expand groups %.1f cycles: records %.0f walk %.0f copy %.0f flush %.0f\n",
                t[6] / d, t[7] / d, t[0] / d, t[1] / d, t[2] / d, t[3] / d, t[4] / d, t[5] / d, t[20] / d, t[13] / d, t[14] / d, t[12] / d, t[8] / d, t[9] / d,
                t[10] / d, t[11] / d);
        fprintf(stderr, "[rsx split] copy per stream: steps %.1f rounds %.1f cycles: scatter %.0f rounds %.0f gather %.0f\n",
                t[19] / d, t[18] / d, t[15] / d, t[16] / d, t[17] / d);
    }
#endif
    e->timed = true;
    RSX_HIP(e, hipGetLastError());
    if (fmt != RSX_FORMAT_DEFLATE) {
        int kind = fmt == RSX_FORMAT_GZIP ? RSX_CHECKSUM_CRC32 : RSX_CHECKSUM_ADLER32;
        uint32_t init = fmt == RSX_FORMAT_GZIP ? 0u : 1u;
        // failed items have out_sizes 0 and cost nothing; the slot sizes are the caller's word, so the
        // segment grid is derived from the produced sizes on the device (trusted_max = false)
        int r = checksum_launch(e, kind, init, out, out_off, out_sizes, n, total_out, max_out, sums, false);
        if (r != RSX_OK) return r;
        verify_trailer_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, e->stream>>>(
            fmt, in, in_off, (uint32_t)n, body_end, sums, out_sizes, status, in_used);
        RSX_HIP(e, hipGetLastError());
    }
    // (the timed span is the whole call: cutting, both decoder kernels, the checksum pass and the trailer check)
    if (!stats) (void)hipEventRecord(e->ev_end, e->stream);
    return RSX_OK;
}
