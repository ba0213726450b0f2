// deflate.hip -- batch raw-DEFLATE / zlib / gzip compression on gfx950.
//
// GPU counterpart of BatchCompressor::compress_batch (src/batch.rs:20-58) and of
// the per-item compressor it drives.  It keeps the reference's decomposition:
// inputs above 256 KiB are cut into 256 KiB chunks that are compressed
// independently and joined by sync-flush markers (src/compress/mod.rs:699-772,
// :662-681), level -> (search depth, nice length, lazy) as in
// src/compress/mod.rs:543-602 and :624-630, hash-chain match finding as in
// src/compress/matchfinder.rs:754-891, dynamic Huffman blocks with
// length-limited canonical codes (src/compress/huffman_comp.rs:125-155,
// src/compress/mod.rs:1775-1883), zlib / gzip framing as in
// src/compress/mod.rs:2248-2357.  The compressed bytes are NOT the reference's:
// every position of a batch is searched in parallel and the parse is resolved
// afterwards, blocks have a fixed size, and a block that does not pay is stored
// (the reference has no stored fallback above level 0).  The contract is a valid
// stream of the same format that inflates to the input.
//
// deflate_chunks_kernel: one 1024-thread workgroup per CU (148 KB of LDS), persistent over the batch's 256 KiB
// chunks (taken from a counter), matching and parsing only.  Positions are taken 1024 at a time:
//   insert  a lane loads its 20 bytes and the 8 in front of them from the sliding LDS copy of the input, tests
//           distances 1..8 in registers and pushes its position: levels 1-6 into a table of 2048 rows x 12 tagged
//           entries (the newest position wins a slot by an LDS atomic max, so the table does not depend on the
//           order the lanes arrive in), levels 7-12 on the reference's hash chains (heads + 16-bit links in the
//           same LDS, pushed wave after wave so that chains stay ordered);
//   search  candidates nearest first -- the row's entries with the position's tag (1/4/4/6/6/8 of them for
//           levels 1-6) or a chain walk of the reference's depth -- each tested at the one byte it must match to
//           beat the best so far, then compared over 20 bytes in registers, longer matches extended in LDS;
//   parse   greedy / one-step / two-step lazy selection as a pointer chase p -> next(p): every wave squares the
//           pointers of its 64 positions through lane exchanges and follows the other waves' exits (one barrier);
//           token units (16 bits: a literal, or a length unit and a distance unit) are compacted by a prefix sum,
//           staged in LDS and stored as 16-byte pieces; LDS atomics keep the symbol histograms;
//   blocks  end at 64 KiB, at the chunk end and (levels 7-12) where the token statistics change
//           (should_end_block); a closed block leaves a table entry and its histogram.
// Levels 10-12 run the batch loop twice around a cost-based parse over recorded match lists (optimal_parse_piece).
// deflate_emit_kernel: one 512-thread workgroup per chunk, four per CU: per block the Huffman codes
// (make_huffman_code spread over the lanes; level 1: the static code), the block type decision, the header
// and the token bits (prefix sum of bit counts, ORs into an LDS stage, 16-byte stores).
// deflate_plan_kernel / deflate_place_kernel: the item's chunks back to back behind the container header, or
// the item as plain stored blocks when that is smaller; Adler-32 / CRC-32 trailer (checksum.hip), size, status.
//
// Roofline: HBM by design (algorithmic bytes per item = uncompressed_len + compressed_len); in practice the match
// kernel is bound by vector-instruction issue in its search phase and by the barriers of a CU that holds one
// workgroup.
#include "engine.h"

#include <cstdlib>

namespace {

constexpr int kCT = 1024;                       // threads per chunk
constexpr uint32_t kChunk = 256 * 1024;         // src/compress/mod.rs:699-700
// Match finder table (LDS): 2^kRowBits rows of kWays entries.  A position goes to the row its 4
// bytes hash to, into the slot its own number hashes to; an entry is (position + 1) << kTagBits |
// tag with kTagBits more bits of the hash, and the newest position wins a slot (atomic max), so
// the table does not depend on the order the lanes arrive in.  (The reference keeps hash chains,
// matchfinder.rs:754-891: a chain is a serial walk of dependent loads, a row is read at once and
// its tags drop nearly every candidate that is only a hash collision.)
#ifndef RSX_SPLIT_FROM_LEVEL
#define RSX_SPLIT_FROM_LEVEL 7
#endif
constexpr int kRowBits = 11;
constexpr uint32_t kRows = 1u << kRowBits;
constexpr uint32_t kWays = 12;
constexpr uint32_t kTagBits = 13;
constexpr uint32_t kTagMask = (1u << kTagBits) - 1;
static_assert(256 * 1024 + 1 < (1u << (32 - kTagBits)), "position + 1 fits above the tag");
// Levels 7-12 keep the reference's hash chains (every older position with the same hash can be
// reached, which is what deep searches pay for); the tables share the rows' LDS.
constexpr int kHashBits = 13;                   // 2^13 heads (the reference hashes to 15 bits, matchfinder.rs:784)
constexpr uint32_t kNearDist = 8;               // distances 1..kNearDist are checked from registers
constexpr uint32_t kWindow = 32768;
// Blocks end where the statistics of the tokens change (src/compress/mod.rs:278-416) and at
// 64 KiB: the reference's soft limit of 300000 bytes (src/common.rs:69) lies above the chunk
// size, and its test asks for savings that grow with the block, so that past 64 KiB it hardly
// ever ends one; data whose kind changes every 64 KiB loses a tenth of its ratio that way.
constexpr uint32_t kBlockSoft = 65535 - kCT - 300;   // ... and once it holds this many bytes
constexpr uint32_t kMinBlock = 5000;            // src/common.rs:68
constexpr uint32_t kObsPerCheck = 2048;         // src/compress/mod.rs:276
constexpr uint32_t kObsTypes = 14;              // 8 literal classes, 2 length classes, 4 offset classes
constexpr uint32_t kTokCap = 66 * 1024;         // 16-bit token units per block (<= bytes per block)
// Bit output (emit kernel): one round of packing (four token units per lane, at most 28 bits each)
// plus the 16-byte unit the round before left open are staged in LDS and leave it as whole
// 16-byte stores.
constexpr int kET = 512;                        // threads of the emit kernel
constexpr uint32_t kUnitsPerLane = 4;            // token units a lane packs per round
constexpr uint32_t kStageWords = kUnitsPerLane * kET * 28 / 32 + 64;
static_assert(kStageWords / 4 < kET, "one stage unit per lane");
constexpr uint32_t kPairs = 8;                  // recorded matches per position (levels 10-12)
constexpr uint32_t kMaxBlocks = kChunk / kMinBlock + 4;   // blocks per chunk
constexpr uint32_t kSegment = 8192;             // positions per piece of the cost-based parse (at least)
constexpr uint32_t kMaxCuts = kChunk / kSegment + kMaxBlocks;
constexpr uint32_t kScratchPad = 1024;          // chunk scratch = chunk_len + kScratchPad (a stored block costs 5-6 bytes more than its data, a chunk has at most kMaxBlocks)
// LDS copy of the input around the current batch: 32 KiB of history, the batch, and
// the longest match behind it; slid forward kInSlide bytes at a time
#ifndef RSX_IN_SLIDE
#define RSX_IN_SLIDE 8192
#endif
constexpr uint32_t kInSlide = RSX_IN_SLIDE;
constexpr uint32_t kInBuf = kWindow + kInSlide + 2048;

__constant__ uint16_t k_len_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,
                                        15, 17, 19, 23, 27, 31, 35, 43, 51,  59,
                                        67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t k_len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                        2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t k_off_base[30] = {1,    2,    3,    4,    5,    7,    9,    13,   17,   25,
                                        33,   49,   65,   97,   129,  193,  257,  385,  513,  769,
                                        1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385,
                                        24577};
__constant__ uint8_t k_off_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3,  3,  4,  4,  5,  5,  6,
                                        6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t k_precode_perm[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                           11, 4,  12, 3, 13, 2, 14, 1, 15};

struct LevelParams {
    uint32_t max_depth, nice_len, lazy;
    uint32_t optimal;   // levels 10-12: cost-based parse over all recorded matches
    uint32_t chains;    // levels 7-12: hash chains instead of the row table
    uint32_t split;     // block ends also follow the statistics of the tokens
};

// src/compress/mod.rs:543-602 and :624-630
__host__ inline LevelParams level_params(int level)
{
    // levels 1-6: candidates compared per position (the row's tags drop the hash collisions a
    // chain of the reference's depth 2, 6, 12, 16, 16, 35 spends most of its steps on);
    // levels 7-12: chain steps, the reference's numbers
    // (level 1: one probe, the newest entry of the row with the position's tag: HtMatchFinder, matchfinder.rs:1109-1232)
    static const uint32_t depth[13] = {0, 1, 4, 4, 6, 6, 8, 100, 300, 600, 35, 100, 300};
    static const uint32_t nice[13] = {0, 32, 10, 14, 30, 30, 65, 130, 258, 258, 75, 150, 258};
    LevelParams p;
    p.max_depth = depth[level];
    p.nice_len = nice[level];
    p.lazy = level >= 8 ? 2 : level >= 5 ? 1 : 0;   // src/compress/mod.rs:624-630
    p.optimal = level >= 10 ? 1 : 0;
    p.chains = level >= 7 ? 1 : 0;
    p.split = level >= RSX_SPLIT_FROM_LEVEL ? 1 : 0;
    return p;
}

// token: literal = byte value; match = 1<<31 | (dist-1) << 8 | (len-3)
__device__ inline uint32_t tok_match(uint32_t len, uint32_t dist) { return 0x80000000u | ((dist - 1) << 8) | (len - 3); }

__device__ inline uint32_t offset_slot(uint32_t dist)
{
    // src/compress/mod.rs:69-105
    if (dist <= 2) return dist - 1;
    uint32_t d = dist - 1;
    uint32_t l = 31 - __clz(d);
    return 2 * l + ((d >> (l - 1)) & 1);
}

struct __align__(16) ChunkLds {
    uint8_t inbuf[kInBuf + 32];      // input bytes [w0 - phase, w0 - phase + kInBuf) of the chunk
    union {
        uint32_t rows[kRows * kWays];    // (position + 1) << kTagBits | tag, 0 = empty
        struct {
            uint32_t head[1 << kHashBits];   // newest position + 1 per 4-byte hash, 0 = empty
            uint16_t prev[32768];            // distance to the next older position with the same hash
        } chain;
    };
    uint16_t m_len[kCT + 2];         // best match per parse slot: slot 0 = last position of the
    uint16_t m_off[kCT + 2];         //   previous batch, slots 1..kCT = this batch
    alignas(16) uint16_t jump[2][kCT + 2];   // pointer-jumping tables (slot -> slot, kCT+1 = past the end)
    alignas(16) uint16_t tok_tail[8];        // token units behind the last complete 16-byte piece of the token buffer
    uint8_t mark[kCT + 2];
    uint8_t len_slot[260];
    uint32_t litlen_freq[288], off_freq[32];
    uint32_t litlen_code[288], off_code[32];   // bit-reversed codeword
    uint8_t litlen_len[288], off_len[32];
    uint32_t scan_tmp[kCT / 64 * 2];
    // control (written by one lane between barriers)
    uint32_t obs_new[16], obs_old[16];   // block split statistics: observations since / before the last check
    uint32_t n_obs_old;
    uint32_t block_ntok;      // 16-bit token units in the open block
    uint32_t entry;           // chunk position of the next token start
    uint32_t block_start;     // chunk position where the open block starts
    uint32_t next_entry;      // scratch for the parse hand-over
    uint32_t tok_base;        // token units of the chunk in front of the open block
    uint32_t nout;            // blocks of the chunk closed so far (entries of the block table)
    uint32_t huff_scratch[288 + 32];   // (levels 10-12: code lengths of the first pass)
    uint32_t len_counts[2][16];
    uint32_t huff_flag[2];           // an alphabet needs the length-limiting walk
    uint32_t blk_end[kMaxBlocks], nblk;   // block ends found by the first pass (levels 10-12)
    uint32_t cut[kMaxCuts], ncut;         // ends of the pieces of the cost-based parse; bit 31 = also a block end
};

// LDS of the emit kernel: one block's histograms and codes, the header scratch, the output stage
struct __align__(16) EmitLds {
    uint8_t len_slot[260];
    uint32_t litlen_freq[288], off_freq[32];
    uint32_t litlen_code[288], off_code[32];   // bit-reversed codeword
    uint8_t litlen_len[288], off_len[32];
    uint32_t scan_tmp[kET / 64 * 2];
    uint32_t block_ntok;      // 16-bit token units of the block
    uint32_t block_start;     // chunk position where the block starts
    uint64_t out_bits;        // bits written to the chunk output so far
    uint32_t stage_word0;     // output word that stage[0] stands for (a multiple of 4)
    alignas(16) uint32_t stage[kStageWords];   // output words [stage_word0, ...): bits are OR-ed in, then flushed
    uint32_t huff_scratch[288 + 32];
    uint32_t len_counts[2][16];
    uint32_t huff_flag[2];           // an alphabet needs the length-limiting walk
    uint32_t pre_freq[19], pre_code[19];
    uint8_t pre_len[20];
    uint32_t nl, no, np;             // code lengths sent per alphabet
    uint32_t body_bits;
    unsigned long long run_mask[5];  // first lanes of the runs of equal code lengths
    alignas(16) uint32_t pj[640 + 160];   // node depths, then the run-length items of the header
};

// Orders the LDS traffic of one wave: its lanes run in step and the LDS serves a wave's requests in
// order, so this only has to keep the compiler from moving accesses across.
#define WAVE_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// ---- bit output -------------------------------------------------------------------
// Token units (16 bits each, in the workgroup's token buffer): a literal is its byte value, a
// match is a length unit kUnitLen | (len - 3) followed by a distance unit kUnitDist | (dist - 1).
constexpr uint32_t kUnitLen = 0x4000, kUnitDist = 0x8000;

// Any lane: OR `nbits` (<= 57) bits of `value` into the stage at absolute bit position `pos` of
// the chunk output.  The position must lie inside the staged window.
__device__ inline void stage_or(EmitLds &L, uint64_t pos, uint64_t value, uint32_t nbits)
{
    if (nbits == 0) return;
    const uint32_t w = (uint32_t)(pos >> 5) - L.stage_word0, sh = (uint32_t)(pos & 31);
    const uint64_t lo = value << sh;
    atomicOr(&L.stage[w], (uint32_t)lo);
    if (sh + nbits > 32) atomicOr(&L.stage[w + 1], (uint32_t)(lo >> 32));
    if (sh + nbits > 64) atomicOr(&L.stage[w + 2], (uint32_t)(value >> (64 - sh)));
}

// All lanes, behind a barrier that follows the ORs: the complete 16-byte units below bit
// `upto` go to the output with plain stores, the unit that is still open moves to the front of
// the stage and the rest of the stage is cleared.
__device__ inline void stage_flush(EmitLds &L, uint32_t *out_words, uint64_t upto)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t w0 = L.stage_word0;
    const uint32_t full = (uint32_t)(upto >> 7) - (w0 >> 2);   // (< kStageWords / 4 <= kET)
    uint4 *st = reinterpret_cast<uint4 *>(L.stage);
    uint4 tail = make_uint4(0, 0, 0, 0);
#ifndef RSX_NO_OUTSTORE
    if (tid < full) reinterpret_cast<uint4 *>(out_words + w0)[tid] = st[tid];
#endif
    if (tid == 0) tail = st[full];
    WG_SYNC();
    if (tid <= full) st[tid] = tail;   // (zero except on lane 0)
    if (tid == 0) L.stage_word0 = w0 + 4 * full;
    WG_SYNC();
}

// All lanes: `count` bytes of the input from `src` as they are, at the byte-aligned bit
// position L.out_bits (stored blocks).
__device__ inline void stage_bytes(EmitLds &L, uint32_t *out_words, const uint8_t *src, uint32_t count)
{
    const uint32_t tid = threadIdx.x;
    for (uint32_t base = 0; base < count; base += 4 * kET) {
        const uint64_t pos = L.out_bits;   // (bits of the rounds before included)
        const uint32_t at = base + 4 * tid;
        if (at < count) {
            const uint32_t nb = count - at < 4 ? count - at : 4;
            uint32_t v = 0;
            for (uint32_t k = 0; k < nb; k++) v |= (uint32_t)src[at + k] << (8 * k);
            stage_or(L, pos + 32ull * tid, v, 8 * nb);
        }
        WG_SYNC();
        const uint32_t done = count - base < 4 * kET ? count - base : 4 * kET;
        if (tid == 0) L.out_bits = pos + 8ull * done;
        stage_flush(L, out_words, pos + 8ull * done);
    }
}

// ---- Huffman code construction (src/compress/huffman_comp.rs) spread over the workgroup ----
// Sorting, node depths, length assignment and codeword numbering are data parallel; the
// two-queue tree build (huffman_comp.rs:38-62) stays on one lane per alphabet, and so does
// the length-limiting walk (huffman_comp.rs:64-89) in the rare case a code exceeds max_len.

constexpr uint32_t kSymBits = 10;
constexpr uint32_t kSymMask = (1u << kSymBits) - 1;

// LDS words behind a non-inlined call: a plain pointer would make every access there a
// flat one
typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ inline uint32_t reverse_bits(uint32_t c, uint32_t len) { return __brev(c) >> (32 - len); }

// one lane: a[0..used) sorted (freq << 10 | sym) -> a[node] = parent << 10 | sym for the
// internal nodes 0..used-3 (node used-2 is the root).  The heads of the leaf queue and of
// the node queue are cached in registers, so an iteration costs one round of LDS reads.
__device__ __noinline__ void huff_build_tree(lds_u32 *a, uint32_t used)
{
    const uint32_t last = used - 1;
    uint32_t i = 0, b = 0, e = 0;
    uint32_t li0 = a[0] >> kSymBits, li1 = a[1] >> kSymBits;   // freq of leaves i, i + 1
    uint32_t wb0 = 0, wb1 = 0;                                 // words of nodes b, b + 1
    while (e < last) {
        uint32_t nf;
        const uint32_t we = a[e];
        if (i < last && (b == e || li1 <= (wb0 >> kSymBits))) {
            nf = li0 + li1;
            i += 2;
            li0 = i <= last ? a[i] >> kSymBits : 0;
            li1 = i + 1 <= last ? a[i + 1] >> kSymBits : 0;
        } else if (b + 2 <= e && (i > last || (wb1 >> kSymBits) < li0)) {
            nf = (wb0 >> kSymBits) + (wb1 >> kSymBits);
            a[b] = (e << kSymBits) | (wb0 & kSymMask);
            a[b + 1] = (e << kSymBits) | (wb1 & kSymMask);
            b += 2;
            wb0 = b < e ? a[b] : 0;
            wb1 = b + 1 < e ? a[b + 1] : 0;
        } else {
            nf = li0 + (wb0 >> kSymBits);
            a[b] = (e << kSymBits) | (wb0 & kSymMask);
            i += 1;
            b += 1;
            li0 = li1;
            li1 = i + 1 <= last ? a[i + 1] >> kSymBits : 0;
            wb0 = wb1;
            wb1 = b + 1 < e ? a[b + 1] : 0;
        }
        const uint32_t nw = (nf << kSymBits) | (we & kSymMask);
        a[e] = nw;
        if (b == e) wb0 = nw;
        else if (b + 1 == e) wb1 = nw;
        e++;
    }
}

// one lane: the walk of huffman_comp.rs:64-89 with its length limiting
__device__ __noinline__ void huff_limit_lengths(lds_u32 *a, uint32_t used, uint32_t max_len,
                                                lds_u32 *len_counts)
{
    for (uint32_t l = 0; l < 16; l++) len_counts[l] = 0;
    len_counts[1] = 2;
    const uint32_t root = used - 2;
    a[root] &= kSymMask;
    for (int node = (int)root - 1; node >= 0; node--) {
        uint32_t parent = a[node] >> kSymBits;
        uint32_t depth = (a[parent] >> kSymBits) + 1;
        a[node] = (a[node] & kSymMask) | (depth << kSymBits);
        if (depth >= max_len) {
            depth = max_len - 1;
            while (len_counts[depth] == 0) depth--;
        }
        len_counts[depth]--;
        len_counts[depth + 1] += 2;
    }
}

// All lanes call; lanes with `on` hold symbol `sym` of an alphabet described by the other
// arguments (several alphabets may be built by disjoint lane groups of one call; a group
// must not straddle the alphabet's lanes over non-participating ones).  pj = 2 * num_syms
// words of scratch, flag = one word of scratch per alphabet, largest_alphabet = the same
// value on every lane: the most symbols any alphabet of this call has.
__device__ void huff_collective(bool on, uint32_t sym, uint32_t num_syms, uint32_t max_len,
                                const uint32_t *freq, uint8_t *lens, uint32_t *codes, uint32_t *a,
                                uint32_t *len_counts, uint32_t *pj, uint32_t *flag,
                                uint32_t largest_alphabet)
{
    // rank sort on (freq, sym) == the order of huffman_comp.rs:8-36
    uint32_t used = 0;
    if (on) {
        const uint32_t f = freq[sym], key = (f << kSymBits) | sym;
        uint32_t rank = 0;
        for (uint32_t t = 0; t < num_syms; t++) {
            const uint32_t ft = freq[t], kt = (ft << kSymBits) | t;
            used += ft != 0;
            rank += (ft != 0) & (kt < key);
        }
        codes[sym] = 0;
        lens[sym] = 0;
        if (f) a[rank] = key;
        if (sym < 16) len_counts[sym] = sym == 1 ? 2 : 0;
        if (sym == 0) *flag = 0;
    }
    WG_SYNC();
    if (on && sym == 0) {
        if (used < 2) {
            // huffman_comp.rs:133-145: two 1-bit codes so the code is complete
            uint32_t s0 = used ? (a[0] & kSymMask) : 0;
            uint32_t nz = s0 ? s0 : 1;
            codes[0] = 0;
            lens[0] = 1;
            codes[nz] = 1;
            lens[nz] = 1;
        } else {
            huff_build_tree((lds_u32 *)a, used);
        }
    }
    WG_SYNC();
    // depth of the internal nodes by pointer jumping: word = depth << 16 | ancestor
    const bool tree = on && used >= 2;
    const uint32_t root = used - 2;
    const bool node = tree && sym <= root;
    if (node) pj[sym] = sym == root ? root : ((1u << 16) | (a[sym] >> kSymBits));
    WG_SYNC();
    uint32_t cur = 0;
    for (uint32_t span = 1; span < largest_alphabet; span <<= 1) {   // same trip count on all lanes
        if (node) {
            const uint32_t w = pj[cur + sym], up = pj[cur + (w & 0xFFFF)];
            pj[(cur ^ num_syms) + sym] = ((w & 0xFFFF0000u) + (up & 0xFFFF0000u)) | (up & 0xFFFF);
        }
        WG_SYNC();
        cur ^= num_syms;
    }
    // without an over-long code, len_counts is the histogram the walk would produce
    if (node && sym != root) {
        const uint32_t depth = pj[cur + sym] >> 16;
        if (depth >= max_len) {
            atomicOr(flag, 1u);
        } else {
            atomicSub(&len_counts[depth], 1u);
            atomicAdd(&len_counts[depth + 1], 2u);
        }
    }
    WG_SYNC();
    if (tree && sym == 0 && *flag) huff_limit_lengths((lds_u32 *)a, used, max_len, (lds_u32 *)len_counts);
    WG_SYNC();
    // gen_codewords (huffman_comp.rs:95-123): sorted index i takes the longest lengths first
    if (tree && sym < used) {
        uint32_t c = 0, len = max_len;
        for (; len >= 1; len--) {
            c += len_counts[len];
            if (sym < c) break;
        }
        lens[a[sym] & kSymMask] = (uint8_t)len;
    }
    WG_SYNC();
    if (tree) {
        const uint32_t len = lens[sym];
        if (len) {
            uint32_t next = 0;
            for (uint32_t l = 2; l <= len; l++) next = (next + len_counts[l - 1]) << 1;
            for (uint32_t t = 0; t < sym; t++) next += lens[t] == len;
            codes[sym] = reverse_bits(next, len);
        }
    }
    WG_SYNC();
}

#ifndef RSX_DEV_TIMERS
#define RSX_DEV_TIMERS 0
#endif
// development timers (cycles summed over workgroups): slide, insert, search, parse, emit, flush
__device__ unsigned long long g_deflate_t[16];
#define DT_MARK(k) do { if (RSX_DEV_TIMERS && threadIdx.x == 0) { unsigned long long t_ = clock64(); dt[k] += t_ - tq; tq = t_; } } while (0)

// exclusive prefix sum over a workgroup of kThreads lanes (returns the total through *total)
// kIdle: nobody can still be reading the sums of the scan before this one (a barrier lies between), so the
// barrier in front of writing them is left out
template <int kThreads, bool kIdle = false, class LS>
__device__ inline uint32_t wg_exclusive_scan(LS &L, uint32_t v, uint32_t *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (!kIdle) WG_SYNC();
    if (lane == 63) L.scan_tmp[w] = x;
    WG_SYNC();
    uint32_t before = 0, all = 0;
    for (int k = 0; k < kThreads / 64; k++) {
        uint32_t s = L.scan_tmp[k];
        if (k < w) before += s;
        all += s;
    }
    *total = all;
    return before + x - v;
}

// the run-length items of `run` equal code lengths `len` (mod.rs:1790-1839)
template <class F>
__device__ inline void rle_run(uint32_t len, uint32_t run, F &&put)
{
    if (len == 0) {
        while (run >= 11) {
            uint32_t k = run < 138 ? run : 138;
            put(18u, k - 11);
            run -= k;
        }
        if (run >= 3) {
            uint32_t k = run < 10 ? run : 10;
            put(17u, k - 3);
            run -= k;
        }
    } else if (run >= 4) {
        put(len, 0u);
        run--;
        while (run >= 3) {
            uint32_t k = run < 6 ? run : 6;
            put(16u, k - 3);
            run -= k;
        }
    }
    while (run > 0) {
        put(len, 0u);
        run--;
    }
}

__device__ inline uint32_t precode_extra_bits(uint32_t sym) { return sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0; }

// All lanes: Huffman codes of the open block, the block type decision (src/compress/mod.rs:
// 1375-1398), its header (mod.rs:1775-1883), the tokens, and a fresh block state.
// in = chunk input, tok = this workgroup's token buffer.
// static_code: the block is coded with the fixed code of RFC 1951 3.2.6 (level 1, src/compress/mod.rs:1499,
// :1570-1583: no code construction and a header of three bits).
__device__ void flush_block(EmitLds &L, const uint8_t *in, uint32_t block_end, bool is_final,
                            const uint16_t *tok, uint32_t *out_words, bool static_code)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t ntok = L.block_ntok;
    const uint32_t bstart = L.block_start;
    const uint32_t block_bytes = block_end - bstart;
    uint32_t *pj = L.pj;                                               // [0, 640): node depths
    uint16_t *items = reinterpret_cast<uint16_t *>(pj + 640);          // <= 320 run-length items
    if (tid == 0) {
        L.litlen_freq[256] = 1;  // end of block
        L.nl = 257;
        L.no = 1;
        L.np = 4;
        L.body_bits = 0;
    }
    if (tid < 19) L.pre_freq[tid] = 0;
    WG_SYNC();
    unsigned long long fq = RSX_DEV_TIMERS ? clock64() : 0;
    const bool is_ll = tid < 288, is_off = tid >= 320 && tid < 352;
    const uint32_t osym = tid - 320;
    if (static_code) {
        // src/compress/mod.rs:153-234: literals 0-143 8 bits from 0x30, 144-255 9 bits from 0x190, 256-279
        // 7 bits from 0, 280-287 8 bits from 0xC0; offsets 5 bits; codewords stored bit-reversed
        if (is_ll) {
            const uint32_t len = tid < 144 ? 8u : tid < 256 ? 9u : tid < 280 ? 7u : 8u;
            const uint32_t code = tid < 144 ? 0x30u + tid : tid < 256 ? 0x190u + (tid - 144) : tid < 280 ? tid - 256 : 0xC0u + (tid - 280);
            L.litlen_len[tid] = (uint8_t)len;
            L.litlen_code[tid] = reverse_bits(code, len);
        } else if (is_off) {
            L.off_len[osym] = 5;
            L.off_code[osym] = reverse_bits(osym, 5);
        }
        WG_SYNC();
    } else {
        huff_collective(is_ll || is_off, is_ll ? tid : osym, is_ll ? 288u : 32u, is_ll ? 14u : 15u,
                        is_ll ? L.litlen_freq : L.off_freq, is_ll ? L.litlen_len : L.off_len,
                        is_ll ? L.litlen_code : L.off_code, is_ll ? L.huff_scratch : L.huff_scratch + 288,
                        is_ll ? L.len_counts[0] : L.len_counts[1], is_ll ? pj : pj + 576,
                        is_ll ? &L.huff_flag[0] : &L.huff_flag[1], 288u);
    }
    if (RSX_DEV_TIMERS && tid == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_deflate_t[6], t_ - fq); fq = t_; }

    // ---- lengths to send, cost of the body ------------------------------------------
    {
        uint32_t bb = 0;
        if (is_ll) {
            const uint32_t l = L.litlen_len[tid];
            if (l) atomicMax(&L.nl, tid + 1);
            if (tid < 286) bb = L.litlen_freq[tid] * (l + (tid >= 257 ? k_len_extra[tid - 257] : 0));
        } else if (is_off) {
            const uint32_t l = L.off_len[osym];
            if (l) atomicMax(&L.no, osym + 1);
            if (osym < 30) bb = L.off_freq[osym] * (l + k_off_extra[osym]);
        }
        for (int o = 32; o; o >>= 1) bb += __shfl_xor(bb, o);
        if ((tid & 63) == 0 && bb) atomicAdd(&L.body_bits, bb);
    }
    WG_SYNC();
    const uint32_t nl = L.nl, no = L.no, total = nl + no;
    auto len_at = [&](uint32_t i) -> uint32_t { return i < nl ? L.litlen_len[i] : L.off_len[i - nl]; };
    uint32_t nitems = 0, ipos = 0, it = 0, np = 4;
    uint64_t hdr_bits = 3;   // (the static code: BFINAL and BTYPE)
    if (!static_code) {
    // ---- run-length items: the first lane of every run of equal lengths writes them ----
    uint32_t mylen = 0;
    bool start = false;
    if (tid < total) {
        mylen = len_at(tid);
        start = tid == 0 || len_at(tid - 1) != mylen;
    }
    {
        const unsigned long long bal = __ballot(start);
        if ((tid & 63) == 0 && tid < 320) L.run_mask[tid >> 6] = bal;
    }
    WG_SYNC();
    uint32_t run = 0, cnt = 0;
    if (start) {
        uint32_t nxt = total;
        const uint32_t w = tid >> 6;
        const unsigned long long m = (L.run_mask[w] >> (tid & 63)) >> 1;
        if (m) {
            nxt = tid + 1 + (uint32_t)(__ffsll((long long)m) - 1);
        } else {
            for (uint32_t w2 = w + 1; w2 < 5; w2++) {
                const unsigned long long m2 = L.run_mask[w2];
                if (m2) {
                    nxt = w2 * 64 + (uint32_t)(__ffsll((long long)m2) - 1);
                    break;
                }
            }
        }
        run = nxt - tid;
        rle_run(mylen, run, [&](uint32_t, uint32_t) { cnt++; });
    }
    uint32_t ioff = wg_exclusive_scan<kET>(L, cnt, &nitems);
    if (start)
        rle_run(mylen, run, [&](uint32_t sym, uint32_t extra) {
            items[ioff++] = (uint16_t)((sym << 8) | extra);
            atomicAdd(&L.pre_freq[sym], 1u);
        });
    WG_SYNC();
    // ---- precode -----------------------------------------------------------------------
    huff_collective(tid < 19, tid, 19u, 7u, L.pre_freq, L.pre_len, L.pre_code, L.huff_scratch,
                    L.len_counts[0], pj, &L.huff_flag[0], 19u);
    if (tid < 19 && L.pre_len[k_precode_perm[tid]]) atomicMax(&L.np, tid + 1);
    uint32_t ibits = 0;
    if (tid < nitems) {
        it = items[tid];
        ibits = L.pre_len[it >> 8] + precode_extra_bits(it >> 8);
    }
    uint32_t items_bits;
    ipos = wg_exclusive_scan<kET>(L, ibits, &items_bits);
    np = L.np;
    hdr_bits = 3 + 5 + 5 + 4 + 3 * np + items_bits;
    }
    const uint64_t body_bits = L.body_bits;
    const uint64_t pos = L.out_bits;
    // as stored blocks of at most 65535 bytes: the first starts at any bit, the others on a byte
    const uint32_t stored_pieces = block_bytes ? (block_bytes + 65534) / 65535 : 1;
    const uint64_t stored_bits = (((pos + 3 + 7) & ~7ull) - pos) + 32 + 8ull * block_bytes + 40ull * (stored_pieces - 1);
    const bool stored = stored_bits <= hdr_bits + body_bits;
    WG_SYNC();   // everyone has read out_bits
    if (RSX_DEV_TIMERS && tid == 0) { unsigned long long t_ = clock64(); atomicAdd(&g_deflate_t[7], t_ - fq); fq = t_; }
    if (RSX_DEV_TIMERS && tid == 0) { atomicAdd(&g_deflate_t[14], 1ull); if (stored) atomicAdd(&g_deflate_t[15], 1ull); }
    if (stored) {
        // stored blocks (src/compress/mod.rs:1408-1448): header bits, align, LEN, NLEN, bytes
        for (uint32_t piece = 0; piece < stored_pieces; piece++) {
            const uint32_t from = piece * 65535, plen = block_bytes - from < 65535 ? block_bytes - from : 65535;
            const uint64_t q0 = L.out_bits, q = (q0 + 3 + 7) & ~7ull;
            WG_SYNC();   // everyone has read out_bits
            if (tid == 0) {
                stage_or(L, q0, (is_final && piece + 1 == stored_pieces) ? 1 : 0, 3);
                stage_or(L, q, plen | ((~plen & 0xFFFF) << 16), 32);
                L.out_bits = q + 32;
            }
            WG_SYNC();
            stage_flush(L, out_words, q + 32);
            stage_bytes(L, out_words, in + bstart + from, plen);
        }
    } else {
        if (static_code) {
            if (tid == 0) stage_or(L, pos, (is_final ? 1 : 0) | (1 << 1), 3);
        } else {
            if (tid == 0) stage_or(L, pos, (is_final ? 1 : 0) | (2 << 1) | ((nl - 257) << 3) | ((no - 1) << 8) | ((np - 4) << 13), 17);
            if (tid < np) stage_or(L, pos + 17 + 3 * tid, L.pre_len[k_precode_perm[tid]], 3);
        }
        if (tid < nitems) {
            const uint32_t sym = it >> 8, l = L.pre_len[sym];
            stage_or(L, pos + 17 + 3 * np + ipos, L.pre_code[sym] | ((it & 0xFF) << l), l + precode_extra_bits(sym));
        }
        WG_SYNC();
        uint64_t at_bits = pos + hdr_bits;
        stage_flush(L, out_words, at_bits);
        // Rounds of four token units per lane: code bits in registers, a prefix sum of the lanes'
        // bit counts, two ORs of up to 56 bits into the stage, a flush of the complete units.
        // The units of the next round are loaded before this round's barriers.
        auto load_units = [&](uint32_t i0, uint32_t (&u)[kUnitsPerLane]) {
            // (a block's units start anywhere in the chunk's token buffer: 16-bit loads)
#pragma unroll
            for (uint32_t h = 0; h < kUnitsPerLane; h++) u[h] = i0 + h < ntok ? tok[i0 + h] : 0;
        };
        uint32_t next_u[kUnitsPerLane];
        load_units(kUnitsPerLane * tid, next_u);
        for (uint32_t base = 0; base < ntok; base += kUnitsPerLane * kET) {
            const uint32_t i0 = base + kUnitsPerLane * tid;
            uint32_t un[kUnitsPerLane];
#pragma unroll
            for (uint32_t h = 0; h < kUnitsPerLane; h++) un[h] = next_u[h];
            if (base + kUnitsPerLane * kET < ntok) load_units(i0 + kUnitsPerLane * kET, next_u);
            uint64_t val[2] = {0, 0};
            uint32_t nb[2] = {0, 0};
#pragma unroll
            for (uint32_t h = 0; h < kUnitsPerLane; h++) {
                const uint32_t u = un[h];
                if (i0 + h < ntok) {
                    uint32_t code, len;
                    if (u & kUnitDist) {
                        const uint32_t dist = (u & 0x7FFF) + 1, os = offset_slot(dist), l2 = L.off_len[os];
                        code = L.off_code[os] | ((dist - k_off_base[os]) << l2);
                        len = l2 + k_off_extra[os];
                    } else if (u & kUnitLen) {
                        const uint32_t ml = (u & 0xFF) + 3, ls = L.len_slot[ml], l1 = L.litlen_len[257 + ls];
                        code = L.litlen_code[257 + ls] | ((ml - k_len_base[ls]) << l1);
                        len = l1 + k_len_extra[ls];
                    } else {
                        code = L.litlen_code[u];
                        len = L.litlen_len[u];
                    }
                    val[h >> 1] |= (uint64_t)code << nb[h >> 1];
                    nb[h >> 1] += len;
                }
            }
            uint32_t round_bits;
            const uint32_t off = wg_exclusive_scan<kET>(L, nb[0] + nb[1], &round_bits);
            stage_or(L, at_bits + off, val[0], nb[0]);
            stage_or(L, at_bits + off + nb[0], val[1], nb[1]);
            WG_SYNC();
            at_bits += round_bits;
            stage_flush(L, out_words, at_bits);
        }
        if (tid == 0) {
            stage_or(L, at_bits, L.litlen_code[256], L.litlen_len[256]);
            L.out_bits = at_bits + L.litlen_len[256];
        }
        WG_SYNC();
        stage_flush(L, out_words, at_bits + L.litlen_len[256]);
    }
    WG_SYNC();
}

// Bytes at any byte offset of an LDS buffer, read as aligned dwords and shifted into place: an LDS
// load whose address is not a multiple of its size is handled one lane at a time (64 cycles per
// wave instruction against 2-8 for an aligned one, tools/micro/lds_rate.hip).
// `buf` is 4-byte aligned; up to 3 bytes past the last requested byte are read as well.
template <int kWords>
__device__ inline void lds_bytes(const uint8_t *buf, uint32_t at, uint32_t (&w)[kWords])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(buf) + (at >> 2);
    uint32_t d[kWords + 1];
#pragma unroll
    for (int k = 0; k <= kWords; k++) d[k] = q[k];
#pragma unroll
    for (int k = 0; k < kWords; k++) w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], at);
}

// Index of the first non-zero byte of the 20 bytes x0..x4 (little endian), 20 if there is none.
__device__ inline uint32_t first_diff_byte(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4)
{
    // v_ffbl_b32 gives 0xFFFFFFFF for zero; the clamped adds keep that value
    uint32_t f0, f1, f2, f3, f4;
    asm("v_ffbl_b32 %0, %1" : "=v"(f0) : "v"(x0));
    asm("v_ffbl_b32 %0, %1" : "=v"(f1) : "v"(x1));
    asm("v_ffbl_b32 %0, %1" : "=v"(f2) : "v"(x2));
    asm("v_ffbl_b32 %0, %1" : "=v"(f3) : "v"(x3));
    asm("v_ffbl_b32 %0, %1" : "=v"(f4) : "v"(x4));
    asm("v_add_u32_e64 %0, %1, 32 clamp" : "=v"(f1) : "v"(f1));
    asm("v_add_u32_e64 %0, %1, 64 clamp" : "=v"(f2) : "v"(f2));
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(f3) : "v"(f3), "s"(96u));    // (not inline constants)
    asm("v_add_u32_e64 %0, %1, %2 clamp" : "=v"(f4) : "v"(f4), "s"(128u));
    uint32_t bit = f0 < f1 ? f0 : f1;
    bit = bit < f2 ? bit : f2;
    uint32_t bit2 = f3 < f4 ? f3 : f4;
    bit = bit < bit2 ? bit : bit2;
    const uint32_t byte = bit >> 3;
    return byte < 20 ? byte : 20;
}

// length of the common prefix of the LDS bytes at offsets a and b of buf, at most max_len
__device__ inline uint32_t match_length(const uint8_t *buf, uint32_t a, uint32_t b, uint32_t max_len)
{
    uint32_t n = 0;
    while (n < max_len) {
        uint32_t x[2], y[2];
        lds_bytes<2>(buf, a + n, x);
        lds_bytes<2>(buf, b + n, y);
        const uint32_t z0 = x[0] ^ y[0], z1 = x[1] ^ y[1];
        if (z0 | z1) {
            n += z0 ? (uint32_t)(__ffs((int)z0) - 1) / 8 : 4 + (uint32_t)(__ffs((int)z1) - 1) / 8;
            break;
        }
        n += 8;
    }
    return n < max_len ? n : max_len;
}

// ---- levels 10-12: cost-based parse (src/compress/mod.rs:1586-1773) --------------------
// The reference runs a greedy pass for statistics, prices every symbol with the resulting
// code lengths (update_costs, mod.rs:2209-2234) and then takes the cheapest path through
// the matches its binary-tree matchfinder reports.  Here the first pass is the ordinary
// batch loop, which also records for every position up to kPairs matches of increasing
// length (nearest offset first), the pricing is the same, and the cheapest path is found
// per piece of kSegment positions by one wave walking backwards: position i is the cheaper
// of a literal and any prefix of any recorded match.  A second run of the batch loop
// replays the chosen steps into tokens.

constexpr uint32_t kBlkCostStride = 320;   // 288 literal/length + 32 offset code lengths

// All lanes, first pass: close the block at block_end, keeping only its code lengths.
__device__ void record_block(ChunkLds &L, uint32_t block_end, uint8_t *blkcost)
{
    const uint32_t tid = threadIdx.x;
    uint32_t *pj = reinterpret_cast<uint32_t *>(L.jump);
    if (tid == 0) L.litlen_freq[256] = 1;
    WG_SYNC();
    const bool is_ll = tid < 288, is_off = tid >= 320 && tid < 352;
    const uint32_t osym = tid - 320;
    huff_collective(is_ll || is_off, is_ll ? tid : osym, is_ll ? 288u : 32u, is_ll ? 14u : 15u,
                    is_ll ? L.litlen_freq : L.off_freq, is_ll ? L.litlen_len : L.off_len,
                    is_ll ? L.litlen_code : L.off_code, is_ll ? L.huff_scratch : L.huff_scratch + 288,
                    is_ll ? L.len_counts[0] : L.len_counts[1], is_ll ? pj : pj + 576,
                    is_ll ? &L.huff_flag[0] : &L.huff_flag[1], 288u);
    const uint32_t k = L.nblk;
    uint8_t *bc = blkcost + k * kBlkCostStride;
    if (is_ll) bc[tid] = L.litlen_len[tid];
    if (is_off) bc[288 + osym] = L.off_len[osym];
    WG_SYNC();
    for (int i = tid; i < 288; i += kCT) L.litlen_freq[i] = 0;
    if (tid < 32) L.off_freq[tid] = 0;
    if (tid < 16) L.obs_new[tid] = L.obs_old[tid] = 0;
    if (tid == 16) L.n_obs_old = 0;
    if (tid == 0) {
        L.blk_end[k] = block_end;
        L.nblk = k + 1;
        L.cut[L.ncut++] = block_end | 0x80000000u;
        L.block_ntok = 0;
        L.block_start = block_end;
    }
    WG_SYNC();
}

// One wave: cheapest path over positions [s, e) of a block whose code lengths are bc.
// ws = 1024 words of LDS private to the wave.  mp[8 * i ..] = recorded matches of position
// i on entry; mp[8 * i] = the chosen step (length 1 = literal) on return.
__device__ void optimal_parse_piece(const ChunkLds &L, const uint8_t *in, uint32_t *mp, uint32_t s,
                                    uint32_t e, const uint8_t *bc, uint32_t *ws)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t *C = ws;                                          // cost to the end, ring of 512
    uint8_t *litc = reinterpret_cast<uint8_t *>(ws + 512);     // [256]
    uint8_t *lenc = litc + 256;                                // [260], index = match length
    uint8_t *offc = lenc + 260;                                // [32], index = offset slot
    for (uint32_t i = lane; i < 256; i += 64) {
        const uint32_t l = bc[i];
        litc[i] = (uint8_t)(l ? l : 13);
    }
    for (uint32_t l = 3 + lane; l <= 258; l += 64) {
        const uint32_t ls = L.len_slot[l], cl = bc[257 + ls];
        lenc[l] = (uint8_t)((cl ? cl : 13) + k_len_extra[ls]);
    }
    if (lane < 30) {
        const uint32_t cl = bc[288 + lane];
        offc[lane] = (uint8_t)((cl ? cl : 12) + k_off_extra[lane]);
    }
    if (lane == 0) C[e & 511] = 0;
    const uint4 *pp = reinterpret_cast<const uint4 *>(mp);
    uint4 na = pp[2 * (size_t)(e - 1)], nb = pp[2 * (size_t)(e - 1) + 1];
    uint32_t nbyte = in[e - 1];
    for (uint32_t i = e; i-- > s;) {
        uint32_t q[8] = {na.x, na.y, na.z, na.w, nb.x, nb.y, nb.z, nb.w};
        const uint32_t byte = nbyte;
        if (i > s) {   // fetch the next position while this one is priced
            na = pp[2 * (size_t)(i - 1)];
            nb = pp[2 * (size_t)(i - 1) + 1];
            nbyte = in[i - 1];
        }
        uint32_t max_len = 0;
        bool valid = true;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            valid = valid && q[k] != 0;
            q[k] = valid ? q[k] : 0;
            max_len = valid ? (q[k] & 511) : max_len;
        }
        max_len = max_len < e - i ? max_len : e - i;
        uint32_t best = ((litc[byte] + C[(i + 1) & 511]) << 9) | 1;
        for (uint32_t base = 3; base <= max_len; base += 64) {
            const uint32_t l = base + lane;
            if (l <= max_len) {
                uint32_t off1 = 0;
#pragma unroll
                for (int k = 7; k >= 0; k--) off1 = (q[k] & 511) >= l ? (q[k] >> 9) : off1;
                const uint32_t cand = ((lenc[l] + offc[offset_slot(off1 + 1)] + C[(i + l) & 511]) << 9) | l;
                best = cand < best ? cand : best;
            }
        }
        for (int o = 32; o; o >>= 1) {
            const uint32_t y = __shfl_xor(best, o);
            best = y < best ? y : best;
        }
        const uint32_t len = best & 511;
        uint32_t off1 = 0;
#pragma unroll
        for (int k = 7; k >= 0; k--) off1 = (q[k] & 511) >= len ? (q[k] >> 9) : off1;
        if (lane == 0) {
            C[i & 511] = best >> 9;
            mp[8 * (size_t)i] = len | (off1 << 9);
        }
    }
}

// One entry of the block table: where the block ends in the chunk and how many token units it has.
// Its histograms (288 literal/length + 32 offset counts, 16 bits each: a block has fewer than 65536
// bytes) lie in a table of their own.
constexpr uint32_t kHistStride = 320;

// All lanes (match kernel): the open block ends at block_end.  Its histograms and size go to the
// chunk's block table for the emit kernel; a fresh block starts.
__device__ inline void close_block(ChunkLds &L, uint32_t block_end, uint2 *blk_tab, uint16_t *blk_hist)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t k = L.nout, units = L.block_ntok;
    if (tid < 288) blk_hist[k * kHistStride + tid] = (uint16_t)L.litlen_freq[tid];
    else if (tid < 320) blk_hist[k * kHistStride + tid] = (uint16_t)L.off_freq[tid - 288];
    if (tid == 0) blk_tab[k] = make_uint2(block_end, units);
    WG_SYNC();
    for (int i = tid; i < 288; i += kCT) L.litlen_freq[i] = 0;
    if (tid < 32) L.off_freq[tid] = 0;
    if (tid < 16) L.obs_new[tid] = L.obs_old[tid] = 0;
    if (tid == 16) L.n_obs_old = 0;
    if (tid == 0) {
        L.nout = k + 1;
        L.tok_base += units;
        L.block_ntok = 0;
        L.block_start = block_end;
    }
    WG_SYNC();
}

// Copy chunk positions [from, to) into the LDS input buffer with 16-byte units
// aligned to the global address.  LDS index of chunk position p is p - w0 + phase.
__device__ inline void stage_input(ChunkLds &L, const uint8_t *in, uint32_t n, uint32_t w0,
                                   uint32_t phase, uint32_t from, uint32_t to)
{
    if (to > n) to = n;
    if (from >= to) return;
    // unit u covers LDS bytes [16u, 16u+16) = global bytes in - phase + w0 + 16u ...
    const uint32_t u0 = (from - w0 + phase) >> 4, u1 = (to - w0 + phase + 15) >> 4;
    const uint8_t *g = in + w0 - phase;   // 16-byte aligned
    for (uint32_t u = u0 + threadIdx.x; u < u1; u += kCT)
        reinterpret_cast<uint4 *>(L.inbuf)[u] = *reinterpret_cast<const uint4 *>(g + 16 * (size_t)u);
}

// Match and parse one chunk (<= 256 KiB): token units into `tok`, one block table entry and one
// histogram per block.  Returns the number of blocks (uniform over the workgroup).
// mp / blkcost: per-workgroup buffers of the cost-based parse (levels 10-12 only).
// kOptimal: levels 10-12 (cost-based parse); a compile-time switch keeps the recording and
// replay code out of the kernel of the other levels
template <bool kOptimal, bool kChains>
__device__ uint32_t compress_chunk(ChunkLds &L, const uint8_t *in, uint32_t n, LevelParams lp, uint16_t *tok,
                                   uint2 *blk_tab, uint16_t *blk_hist, uint32_t *mp, uint8_t *blkcost)
{
    const int tid = threadIdx.x;
    // reset the tables
    {
        for (uint32_t i = tid; i < (kChains ? (1u << kHashBits) : kRows * kWays) / 4; i += kCT)
            reinterpret_cast<uint4 *>(L.rows)[i] = make_uint4(0, 0, 0, 0);   // (the chain heads lie in front)
        for (int i = tid; i < 288; i += kCT) L.litlen_freq[i] = 0;
        if (tid < 32) L.off_freq[tid] = 0;
        if (tid < 16) L.obs_new[tid] = L.obs_old[tid] = 0;
        if (tid == 16) L.n_obs_old = 0;
        for (int i = tid; i < 260; i += kCT) {
            uint32_t s = 0;
            while (s + 1 < 29 && k_len_base[s + 1] <= (uint32_t)i) s++;
            L.len_slot[i] = (uint8_t)s;
        }
        if (tid == 0) {
            L.block_ntok = 0;
            L.entry = 0;
            L.block_start = 0;
            L.tok_base = 0;
            L.nout = 0;
            L.m_len[0] = 0;
            L.m_off[0] = 0;
            L.nblk = 0;
            L.ncut = 0;
        }
    }
    WG_SYNC();

    // LDS input window
    const uint32_t phase = (uint32_t)((uintptr_t)in & 15);
    uint32_t w0 = 0, staged = kInBuf - 16 < n ? kInBuf - 16 : n;
    if (lp.max_depth != 0) stage_input(L, in, n, 0, phase, 0, staged);
    WG_SYNC();

    if (lp.max_depth == 0) n = 0;   // level 0: stored blocks only, written by the emit kernel

    unsigned long long dt[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tq = RSX_DEV_TIMERS ? clock64() : 0;
    // Levels 10-12 run the batch loop twice: a first pass that searches, records the
    // matches and the block ends, then (after the cost-based parse) a replay pass that
    // turns the chosen steps into tokens block by block.
    const uint32_t passes = kOptimal && n ? 2 : 1;
    for (uint32_t pass = 0; pass < passes; pass++) {
    const bool analysis = kOptimal && pass == 0, replay = kOptimal && pass == 1;
    uint32_t blk = 0;
    uint32_t n_eff = n;   // replay: end of the block being replayed
    if (replay) {
        __threadfence();
        WG_SYNC();
        // pieces of all blocks, dealt round-robin to the waves; the hash tables are idle now
        {
            const uint32_t wave = tid >> 6;
            uint32_t *ws = L.rows + wave * 1024;   // (the match finder's tables are idle by now)
            static_assert(sizeof(L.rows) >= (kCT / 64) * 4096, "per-wave scratch of the parse");
            uint32_t k = 0, s0 = 0;
            for (uint32_t piece = 0; piece < L.ncut; piece++) {
                const uint32_t c = L.cut[piece], e0 = c & 0x7FFFFFFFu;
                if (piece % (kCT / 64) == wave && e0 > s0)
                    optimal_parse_piece(L, in, mp, s0, e0, blkcost + k * kBlkCostStride, ws);
                s0 = e0;
                k += c >> 31;
            }
        }
        __threadfence();
        WG_SYNC();
        if (tid == 0) {
            L.entry = 0;
            L.block_start = 0;
            L.block_ntok = 0;
            L.m_len[0] = 0;
            L.m_off[0] = 0;
        }
        WG_SYNC();
        n_eff = L.blk_end[0];
    }
    for (uint32_t b0 = 0; b0 < n;) {
        // ---- slide the LDS input window when the batch (plus the longest match) would
        //      run off its end; 32 KiB of history stay behind b0 ---------------------------
        if (!replay && b0 + kCT + 258 + 8 > w0 + kInBuf - 16 && staged < n) {
            constexpr uint32_t kUnits = (kInBuf - kInSlide) / 16;
            static_assert(kUnits <= 3 * kCT, "three 16-byte units per lane");
            const uint4 *from = reinterpret_cast<const uint4 *>(L.inbuf) + kInSlide / 16;
            uint4 *to = reinterpret_cast<uint4 *>(L.inbuf);
            const uint32_t u0 = tid, u1 = tid + kCT, u2 = tid + 2 * kCT;
            // (named registers: an indexed array of three ends up in scratch memory here)
            uint4 va = from[u0], vb = from[u1], vc = make_uint4(0, 0, 0, 0);
            if (u2 < kUnits) vc = from[u2];
            WG_SYNC();
            to[u0] = va;
            to[u1] = vb;
            if (u2 < kUnits) to[u2] = vc;
            w0 += kInSlide;
            WG_SYNC();
            const uint32_t upto = w0 + kInBuf - 16 < n ? w0 + kInBuf - 16 : n;
            // restage from the 16-byte unit that holds `staged` (its head is already there)
            stage_input(L, in, n, w0, phase, staged, upto);
            staged = upto;
            WG_SYNC();
        }
        DT_MARK(0);
        // LDS address of the input byte at chunk position q (q >= w0)
        auto inp = [&](uint32_t q) -> const uint8_t * { return &L.inbuf[q + phase - w0]; };
        auto byte_at = [&](uint32_t q) -> uint32_t { return replay ? in[q] : *inp(q); };
        // ---- match: slot i+1 of the parse tables = position b0 + i ----------------
        const uint32_t p = b0 + tid;
        uint32_t best_len = 0, best_off = 0;
        const bool can = !replay && p + 3 <= n;   // matchfinder.rs:765-767
        const bool can4 = can && p + 4 <= n;
        uint32_t src4 = 0, row = 0, tag = 0, near = 0, w0r = 0, w1r = 0;
        uint64_t src_a = 0, src_b = 0;
        bool repeats = false;
        if (can) {
            {
                uint32_t w[5];
                lds_bytes<5>(L.inbuf, p + phase - w0, w);   // bytes past the chunk end are never compared
                src4 = w[0];
                src_a = w[1] | ((uint64_t)w[2] << 32);
                src_b = w[3] | ((uint64_t)w[4] << 32);
            }
            const uint32_t h = src4 * 0x1E35A7BDu;   // multiplier of matchfinder.rs:784
            row = h >> (32 - kRowBits);
            tag = (h >> (32 - kRowBits - kTagBits)) & kTagMask;
            // distances 1..8: the bytes there are in registers with the 8 bytes in front of p
            const uint32_t back = p < kNearDist ? p : kNearDist;
            uint64_t before = 0;
            if (back) {
                uint32_t w[2];
                lds_bytes<2>(L.inbuf, p - back + phase - w0, w);
                before = (w[0] | ((uint64_t)w[1] << 32)) << (8 * (kNearDist - back));   // byte 7 = position p - 1
            }
            w0r = (uint32_t)before;
            w1r = (uint32_t)(before >> 32);
#pragma unroll
            for (uint32_t d = 1; d <= kNearDist; d++) {
                const uint32_t o = kNearDist - d;   // byte offset of position p - d in (before, src4)
                const uint32_t c4 = o == 0 ? w0r : o < 4 ? __builtin_amdgcn_alignbyte(w1r, w0r, o)
                                  : o == 4 ? w1r : __builtin_amdgcn_alignbyte(src4, w1r, o - 4);
                const uint32_t x = c4 ^ src4;
                if ((x & 0xFFFFFF) == 0 && d <= back) near |= 1u << d;
                repeats = repeats || (x == 0 && d <= back);
            }
        }
        uint32_t link = 0;
        if (kChains) {
            // push on the chains one wave after the other, so that within a chain newer
            // positions always sit in front of older ones (matchfinder.rs:788-805).  Heads hold
            // position + 1, links the distance to the next older position (0 ends the chain).
            const uint32_t h = (src4 * 0x1E35A7BDu) >> (32 - kHashBits);
            for (int w = 0; w < kCT / 64; w++) {
                if ((tid >> 6) == w && can) {
                    if (can4) {
                        const uint32_t o = atomicExch(&L.chain.head[h], p + 1), dlt = p + 1 - o;
                        link = (o != 0 && dlt < 65536) ? dlt : 0;
                    }
                    L.chain.prev[p & 32767] = (uint16_t)link;
                }
                if (w + 1 < kCT / 64) WG_SYNC();
            }
        } else if (can4 && !repeats) {
            // Push the position on its row.  A position whose 4 bytes repeat at a distance of
            // 1..8 stays out (runs and short periods would flood one row, and the near check
            // finds them).
            const uint32_t slot = (((p * 0x9E3779B1u) >> 16) * kWays) >> 16;
            atomicMax(&L.rows[row * kWays + slot], ((p + 1) << kTagBits) | tag);
        }
        WG_SYNC();
        DT_MARK(1);
        if (can && lp.max_depth) {
            // Candidates (matchfinder.rs:808-870 walks a chain here): first the near distances that
            // passed the three-byte test, then the row's entries with this position's tag, newest
            // first, two per round so that their loads overlap.  A candidate costs one round of
            // LDS reads (20 bytes) compared with the position's own first 20 bytes in registers;
            // only longer matches go through the extension loop.
            const uint32_t max_len = n - p < 258 ? n - p : 258;
            const uint32_t stop_len = lp.nice_len < max_len ? lp.nice_len : max_len;
            const uint32_t max_dist = p < 32768 ? p : 32768;   // matchfinder.rs:818
            const uint32_t src_at = p + phase - w0;   // offset of position p in the LDS input window
            uint32_t npairs = 0;
            uint32_t *mpp = mp + (size_t)p * kPairs;
            const uint32_t S[5] = {src4, (uint32_t)src_a, (uint32_t)(src_a >> 32), (uint32_t)src_b, (uint32_t)(src_b >> 32)};
            // length of the match at `dist` given its first 20 bytes
            auto finish_length = [&](const uint32_t (&w)[5], uint32_t dist) -> uint32_t {
                uint32_t l = first_diff_byte(w[0] ^ S[0], w[1] ^ S[1], w[2] ^ S[2], w[3] ^ S[3], w[4] ^ S[4]);
                if (l == 20 && max_len > 20)
                    l = 20 + match_length(L.inbuf, src_at - dist + 20, src_at + 20, max_len - 20);
                l = l < max_len ? l : max_len;         // bytes past the chunk end do not count
                return l < 3 ? 0u : l;   // (matchfinder.rs:835-853: nothing shorter than three bytes counts)
            };
            auto consider = [&](uint32_t l, uint32_t dist) {
                const bool better = l > best_len;   // candidates come nearest first
                if (analysis && l > best_len) {
                    // matches of increasing length; the last slot keeps the longest
                    const uint32_t slot = npairs < kPairs ? npairs : kPairs - 1;
                    mpp[slot] = l | ((dist - 1) << 9);
                    npairs = slot + 1;
                }
                best_len = better ? l : best_len;
                best_off = better ? dist : best_off;
            };
            uint32_t it_near = 0, it_row = 0, it_full = 0;
            unsigned long long tn0 = RSX_DEV_TIMERS ? clock64() : 0;
            // near distances: the candidate's first 20 bytes are the 8 bytes in front of p followed
            // by p's own, all in registers
#ifdef RSX_SKIP_NEAR
            near = 0;
#endif
            {
                // (a loop over the set bits, nearest first: an unrolled test of every distance would
                // run all eight bodies whenever any lane of the wave has that bit)
                const uint32_t C[7] = {w0r, w1r, S[0], S[1], S[2], S[3], S[4]};
                while (near && best_len < stop_len) {
                    const uint32_t d = (uint32_t)__ffs((int)near) - 1;
                    near &= near - 1;
                    it_near++;
                    // a candidate that differs from p at offset best_len cannot be longer
                    if (best_len && L.inbuf[src_at - d + best_len] != L.inbuf[src_at + best_len]) continue;
                    const uint32_t o = kNearDist - d;   // byte offset of position p - d in C
                    const bool up = o >= 4;
                    uint32_t w[5];
#pragma unroll
                    for (uint32_t k = 0; k < 5; k++)
                        w[k] = __builtin_amdgcn_alignbyte(up ? C[k + 2] : C[k + 1], up ? C[k + 1] : C[k], o);
                    consider(finish_length(w, d), d);
                }
            }
            unsigned long long tn1 = RSX_DEV_TIMERS ? clock64() : 0;
            if (kChains) {
                // Walk the chain (matchfinder.rs:808-870), nearest first.  A candidate that differs
                // from p at offset best_len cannot be longer than the best so far: one byte decides.
                uint32_t dist = 0, depth = 0;
                uint32_t next_byte = best_len ? L.inbuf[src_at + best_len] : 0;
                bool go = link != 0 && link <= max_dist && best_len < stop_len;
                while (go) {
                    dist += link;
                    link = L.chain.prev[(p - dist) & 32767];
                    if (!best_len || L.inbuf[src_at - dist + best_len] == next_byte) {
                        uint32_t w[5];
                        lds_bytes<5>(L.inbuf, src_at - dist, w);
                        const uint32_t before_len = best_len;
                        consider(finish_length(w, dist), dist);
                        if (best_len != before_len && best_len < max_len) next_byte = L.inbuf[src_at + best_len];
                    }
                    depth++;
                    it_row++;
                    // matchfinder.rs:825, :851: stop at a nice match, at the end of the chain, of the
                    // window, of the search budget
                    go = best_len < stop_len && link != 0 && dist + link <= max_dist && depth < lp.max_depth;
                }
#ifdef RSX_SKIP_ROWS
            } else if (false) {
#else
            } else if (can4 && best_len < stop_len) {
#endif
                const uint4 *rp = reinterpret_cast<const uint4 *>(&L.rows[row * kWays]);
                const uint4 ra = rp[0], rb = rp[1], rc = rp[2];
                const uint32_t e[kWays] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w, rc.x, rc.y, rc.z, rc.w};
                static_assert(kWays == 12, "three 16-byte reads per row");
                // candidates: entries with this position's tag that lie in front of it inside the
                // window (another tag, an empty entry, this or a later position of the batch fall
                // out), taken nearest first.  (p + 1) << kTagBits | tag minus an entry with the same
                // tag is distance << kTagBits; rotated right by kTagBits it is the distance, and
                // with another tag a number far above any distance.  key = distance, ~0 = none.
                const uint32_t mine = ((p + 1) << kTagBits) | tag;
                // keys: distance - 1 as 16 bits, two to a register, 0xFFFF = none
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                u16x2 key[kWays / 2];
#pragma unroll
                for (uint32_t k = 0; k < kWays; k += 2) {
                    uint32_t k16[2];
#pragma unroll
                    for (uint32_t h = 0; h < 2; h++) {
                        const uint32_t t = mine - e[k + h];
                        const uint32_t d = __builtin_amdgcn_alignbit(t, t, kTagBits);
                        k16[h] = d - 1 < max_dist ? d - 1 : 0xFFFFu;
                    }
                    key[k / 2] = __builtin_bit_cast(u16x2, k16[0] | (k16[1] << 16));
                }
                uint32_t depth = 0;
                // Nearest first: the smallest key not yet taken.  Keys minus (last taken + 1), modulo
                // 2^16, keep the order of the keys that are left and push the taken ones above them.
                uint32_t taken = 0;   // last taken key + 1
                while (depth < lp.max_depth && best_len < stop_len) {
                    const u16x2 base2 = __builtin_bit_cast(u16x2, taken * 0x10001u);
                    u16x2 m = key[0] - base2;
#pragma unroll
                    for (uint32_t k = 1; k < kWays / 2; k++) m = __builtin_elementwise_min(m, (u16x2)(key[k] - base2));
                    const uint32_t mm = __builtin_bit_cast(uint32_t, m);
                    const uint32_t r = (mm & 0xFFFF) < (mm >> 16) ? (mm & 0xFFFF) : (mm >> 16);
                    if (r + taken >= 32768) break;   // what is left is taken or none
                    const uint32_t d0 = r + taken + 1;
                    taken = d0;
                    depth++;
                    it_row++;
                    // (testing first the one byte that a longer match must have costs an LDS round trip of
                    // its own: 159.8 ms per batch of 4096 x 1 MiB with the test, 156.5 without)
                    uint32_t w[5];
                    it_full++;
                    lds_bytes<5>(L.inbuf, src_at - d0, w);
                    consider(finish_length(w, d0), d0);
                }
            }
            if (RSX_DEV_TIMERS) {
                unsigned long long tn2 = clock64();
                uint32_t sum_row = it_row, sum_full = it_full;
                for (int o = 32; o; o >>= 1) {
                    it_near = max(it_near, (uint32_t)__shfl_xor((int)it_near, o));
                    it_row = max(it_row, (uint32_t)__shfl_xor((int)it_row, o));
                    sum_row += (uint32_t)__shfl_xor((int)sum_row, o);
                    sum_full += (uint32_t)__shfl_xor((int)sum_full, o);
                }
                if (tid == 0) {
                    atomicAdd(&g_deflate_t[12], (unsigned long long)sum_row);
                    atomicAdd(&g_deflate_t[13], (unsigned long long)sum_full);
                }
                if (tid == 0) {
                    atomicAdd(&g_deflate_t[8], tn1 - tn0);
                    atomicAdd(&g_deflate_t[9], tn2 - tn1);
                    atomicAdd(&g_deflate_t[10], (unsigned long long)it_near);
                    atomicAdd(&g_deflate_t[11], (unsigned long long)it_row);
                }
            }
            if (analysis && npairs < kPairs) mpp[npairs] = 0;
        } else if (analysis && p < n) {
            mp[(size_t)p * kPairs] = 0;
        } else if (replay && p < n_eff) {
            const uint32_t w = mp[(size_t)p * kPairs], len = w & 511;
            best_len = len >= 3 ? len : 0;
            best_off = (w >> 9) + 1;
        }
        L.m_len[tid + 1] = (uint16_t)best_len;
        L.m_off[tid + 1] = (uint16_t)(best_off - (best_off ? 1 : 0));  // dist-1 fits 15 bits
        WG_SYNC();
        DT_MARK(2);

        // ---- parse: slots [0, lim) are decided in this batch ----------------------
        // slot s <-> position b0 - 1 + s.  The last position of the batch waits for its
        // successor's match (lazy rule) unless the chunk ends here.
        const bool tail = b0 + kCT >= n_eff;
        const uint32_t nslots = (tail ? n_eff - b0 : kCT) + 1;    // slots that hold a position
        const uint32_t lim = tail ? nslots : nslots - 1;      // slots decided now
        const uint32_t base_pos = b0 - 1;                     // position of slot 0 (b0 = 0: unused)
        const uint32_t entry = L.entry;
        const uint32_t kEnd = kCT + 1;
        // next(s): slot of the following token start, tokens emitted by a start at s
        uint32_t my_next = kEnd, ntoks = 0, t0 = 0, t1 = 0, t2 = 0;
        {
            const uint32_t s = tid;
            if (s < lim && !(b0 == 0 && s == 0)) {
                uint32_t len = L.m_len[s], pos = base_pos + s;
                if (len >= 3) {
                    uint32_t len1 = (s + 1 < nslots) ? L.m_len[s + 1] : 0;
                    if (lp.lazy && !replay && len < lp.nice_len && len1 > len) {
                        // compress/mod.rs:1290-1336: a literal, then the longer match; with two lazy
                        // steps (levels 8, 9) two literals if the position after that is longer still
                        // (at the end of a batch, where that position is not searched yet, one step)
                        const uint32_t len2 = (lp.lazy >= 2 && s + 2 < nslots) ? L.m_len[s + 2] : 0;
                        t0 = byte_at(pos);
                        if (len2 > len1) {
                            t1 = byte_at(pos + 1);
                            t2 = tok_match(len2, (uint32_t)L.m_off[s + 2] + 1);
                            ntoks = 3;
                            my_next = s + 2 + len2;
                        } else {
                            t1 = tok_match(len1, (uint32_t)L.m_off[s + 1] + 1);
                            ntoks = 2;
                            my_next = s + 1 + len1;
                        }
                    } else {
                        t0 = tok_match(len, (uint32_t)L.m_off[s] + 1);
                        ntoks = 1;
                        my_next = s + len;
                    }
                } else {
                    t0 = byte_at(pos);
                    ntoks = 1;
                    my_next = s + 1;
                }
            }
        }
        // the last lane also covers slot kCT when the chunk ends inside this batch
        // (nslots = kCT + 1 only when tail and n - b0 == kCT): handled below by lane kCT-1
        uint32_t x_next = kEnd, x_t0 = 0;
        if (tid == kCT - 1 && lim == kCT + 1) {
            const uint32_t s = kCT;
            uint32_t len = L.m_len[s], pos = base_pos + s;
            if (len >= 3) {
                x_t0 = tok_match(len, (uint32_t)L.m_off[s] + 1);
                x_next = s + len;
            } else {
                x_t0 = byte_at(pos);
                x_next = s + 1;
            }
        }
        // Which slots start a token: the chain entry -> next(entry) -> ... over slots 0..kCT (slot kCT+1 =
        // beyond).  Every wave first squares the pointers of its own 64 slots through lane exchanges, six times,
        // which needs no barrier: that leaves, per slot, the slot at which its chain leaves the wave's segment,
        // and on the way the 2^k-th successors inside the segment.  The exits go to LDS (the one barrier of the
        // phase); every wave follows them from the batch's entry to its own segment, then marks its part of
        // the chain: marked slots mark their 2^k-th successor, k = 0..5, through the wave's own bytes of
        // L.mark.  (Pointer jumping over the whole batch took nine rounds with a barrier each, on a CU that
        // holds one workgroup.)
        const uint32_t entry_slot = entry - base_pos;   // entry >= base_pos (b0 = 0: entry 0 -> slot 1)
        {
            const uint32_t lane = tid & 63, seg0 = (uint32_t)tid & ~63u, seg1 = seg0 + 64;
            uint32_t jk[6];   // 2^k-th successor (a slot of a later segment stays as it is)
            uint32_t j = my_next > kEnd ? kEnd : my_next;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                jk[k] = j;
                const bool inside = j < seg1;
                const uint32_t t = (uint32_t)__shfl((int)j, inside ? (int)(j - seg0) : (int)lane);
                j = inside ? t : j;
            }
            L.jump[0][tid] = (uint16_t)j;   // where the chain through this slot leaves the segment
            WG_SYNC();
            uint32_t e = entry_slot;   // (the same for every lane of the wave)
            while (e < seg0 && e < lim && e < (uint32_t)kCT) e = L.jump[0][e];
            const uint32_t mine = (e >= seg0 && e < seg1 && e < lim) ? e : ~0u;
            if (tid == kCT - 1) {
                // slot kCT (the chunk ends with this batch) belongs to no segment: the chain reaches it or not
                const uint32_t out = mine != ~0u ? (uint32_t)L.jump[0][mine] : e;
                L.mark[kCT] = out == (uint32_t)kCT && (uint32_t)kCT < lim;
                L.mark[kEnd] = 0;
            }
            bool marked = (uint32_t)tid == mine;
            L.mark[tid] = marked;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                WAVE_SYNC();
                if (marked && jk[k] < seg1 && jk[k] < lim) L.mark[jk[k]] = 1;
                WAVE_SYNC();
                marked = L.mark[tid] != 0;
            }
            WAVE_SYNC();
        }
        DT_MARK(3);
        // emit the tokens of marked slots in position order
        const bool live = tid < lim && L.mark[tid] != 0;
        const bool livex = tid == kCT - 1 && lim == kCT + 1 && L.mark[kCT] != 0;
        // 16-bit units: one per literal, two per match
        auto units_of = [](uint32_t t) -> uint32_t { return t >> 31 ? 2u : 1u; };
        // The batch's units are collected in LDS (the pointer-jumping tables are idle by now) and
        // leave it as whole 16-byte pieces: two-byte stores scattered over the lanes cost several
        // times their size in write traffic.  The stage starts with the units the batch before
        // left behind its last complete piece, so a piece of the stage is a piece of the buffer.
        // A position holds at most one unit on average (a literal: one unit per byte, a match:
        // two per three bytes or more).
        uint16_t *ustage = reinterpret_cast<uint16_t *>(L.jump);
        static_assert(sizeof(L.jump) >= (kCT + 32) * 2, "one batch of token units");
        auto put_units = [&](uint32_t at, uint32_t t) {
            if (t >> 31) {
                ustage[at] = (uint16_t)(kUnitLen | (t & 0xFF));
                ustage[at + 1] = (uint16_t)(kUnitDist | ((t >> 8) & 0x7FFF));
            } else {
                ustage[at] = (uint16_t)t;
            }
        };
        // histograms (compress/mod.rs:1306-1351)
        auto count_token = [&](uint32_t t) {
            if (t >> 31) {
                const uint32_t len = (t & 0xFF) + 3, os = offset_slot(((t >> 8) & 0x7FFF) + 1);
                atomicAdd(&L.litlen_freq[257 + L.len_slot[len]], 1u);
                atomicAdd(&L.off_freq[os], 1u);
            } else {
                atomicAdd(&L.litlen_freq[t], 1u);
            }
        };
        const uint32_t u0 = units_of(t0), u1 = ntoks >= 2 ? units_of(t1) : 0, u2 = ntoks >= 3 ? units_of(t2) : 0;
        const uint32_t my_units = live ? u0 + u1 + u2 : 0;
        uint32_t cnt = my_units + (livex ? units_of(x_t0) : 0);
        uint32_t total;
        // (the scan before this one was the batch before's: barriers lie between)
        uint32_t at = wg_exclusive_scan<kCT, true>(L, cnt, &total);
        const uint32_t tok_base = L.tok_base, tbase = tok_base + L.block_ntok, ta = tbase & 7;   // units of the chunk so far
        if ((uint32_t)tid < ta) ustage[tid] = L.tok_tail[tid];
        if (live) {
            put_units(ta + at, t0);
            count_token(t0);
            if (ntoks >= 2) {
                put_units(ta + at + u0, t1);
                count_token(t1);
            }
            if (ntoks >= 3) {
                put_units(ta + at + u0 + u1, t2);
                count_token(t2);
            }
            // the marked slot whose successor leaves the decided range hands over
            if (my_next >= lim) L.next_entry = base_pos + my_next;
        }
        if (livex) {
            put_units(ta + at + my_units, x_t0);
            count_token(x_t0);
            L.next_entry = base_pos + x_next;
        }
        WG_SYNC();
        {
            // pieces of the stage -> tok[tbase - ta ...], 16 bytes per lane (the last one may end in
            // units that are not there yet: the next batch writes that piece again)
            const uint32_t have = ta + total, pieces = (have + 7) >> 3;   // (<= (kCT + 32) / 8)
            if ((uint32_t)tid < pieces)
                reinterpret_cast<uint4 *>(tok + (tbase - ta))[tid] = reinterpret_cast<const uint4 *>(ustage)[tid];
            if ((uint32_t)tid < (have & 7)) L.tok_tail[tid] = ustage[(have & ~7u) + tid];
        }
        // The batch's results: where the next token starts and the units of the open block.  Every lane works
        // them out for itself (the hand-over slot was written in front of the barrier above); the first lane
        // also leaves them in LDS for the next batch, which reads them behind barriers of its own, so no
        // barrier follows here.
        const uint32_t covered = total ? L.next_entry : entry;   // every byte below is inside an emitted token
        const uint32_t open_ntok = tbase - tok_base + total;
        if (tid == 0) {
            L.entry = covered;
            L.block_ntok = open_ntok;
            // carry the last position's match to slot 0 of the next batch
            L.m_len[0] = L.m_len[kCT];
            L.m_off[0] = L.m_off[kCT];
        }
        DT_MARK(4);
        // ---- close the block when it is large enough or the chunk ends -------------
        if (replay) {
            if (tail) {
                WG_SYNC();   // (close_block reads what the first lane has just written)
                close_block(L, n_eff, blk_tab, blk_hist);
                b0 = n_eff;
                blk++;
                n_eff = blk < L.nblk ? L.blk_end[blk] : n;
            } else {
                b0 += kCT;
            }
        } else {
            // should_end_block (compress/mod.rs:387-415), looked at once per batch: with enough new
            // observations in a block of at least kMinBlock bytes that does not end the chunk,
            // the block ends if coding the new observations with their own statistics instead
            // of the block's saves more than a sixteenth of a bit per byte of the block
            // (do_end_block_check, :359-384); otherwise they join the block's.
            // The observation classes (:303-346: literals by their top three bits, lengths below
            // and from 8, offsets up to 256, 4096 and 32768) are sums over the block's symbol
            // histograms, and the number of observations is the number of token units.
            bool split = false;
            {
                const uint32_t n_old = L.n_obs_old, n_new = open_ntok - n_old;
                const uint32_t block_len = covered - L.block_start;
                if (lp.split && !tail && n_new >= kObsPerCheck && n - covered > kMinBlock && block_len >= kMinBlock) {
                    if (tid < kObsTypes) {
                        const uint32_t *f = tid < 10 ? L.litlen_freq : L.off_freq;
                        const uint32_t lo = tid < 8 ? 32 * tid : tid == 8 ? 257 : tid == 9 ? 262 : tid == 10 ? 0 : tid == 11 ? 16 : 24;
                        const uint32_t hi = tid < 8 ? lo + 32 : tid == 8 ? 262 : tid == 9 ? 286 : tid == 10 ? 16 : tid == 11 ? 24 : tid == 12 ? 30 : 24;
                        uint32_t sum = 0;
                        for (uint32_t k = lo; k < hi; k++) sum += f[k];
                        L.obs_new[tid] = sum - L.obs_old[tid];
                    }
                    WG_SYNC();
                    if (n_old) {
                        const uint32_t lg_old = 31 - __clz(n_old), lg_new = 31 - __clz(n_new);
                        uint32_t old_bits = 0, new_bits = 0;
#pragma unroll
                        for (uint32_t i = 0; i < kObsTypes; i++) {
                            const uint32_t nw = L.obs_new[i];
                            const uint32_t lo = 31 - __clz(L.obs_old[i] + 1), ln = 31 - __clz(nw + 1);
                            old_bits += nw * (lg_old > lo ? lg_old - lo : 0);
                            new_bits += nw * (lg_new > ln ? lg_new - ln : 0);
                        }
                        split = (int32_t)(old_bits - new_bits) > (int32_t)(block_len / 16);
                    }
                    if (!split) {
                        WG_SYNC();   // everyone has read the statistics
                        if (tid < kObsTypes) L.obs_old[tid] += L.obs_new[tid];
                        if (tid == 16) L.n_obs_old = n_old + n_new;
                    }
                }
            }
            if (tail || split || covered - L.block_start >= kBlockSoft || open_ntok + 3 * kCT + 4 > kTokCap) {
                const uint32_t block_end = tail ? n : covered;
                WG_SYNC();   // (both read what the first lane has just written)
                if (analysis) record_block(L, block_end, blkcost);
                else close_block(L, block_end, blk_tab, blk_hist);
            } else if (analysis) {
                // the parse is cut into pieces where the first pass has a token boundary, so
                // the cheapest path of a piece is never dearer than the first pass's tokens
                if (tid == 0) {
                    const uint32_t last = L.ncut ? (L.cut[L.ncut - 1] & 0x7FFFFFFFu) : 0;
                    if (covered - last >= kSegment) L.cut[L.ncut++] = covered;
                }
                WG_SYNC();
            }
            b0 += kCT;
        }
        DT_MARK(5);
    }
    }
    if (RSX_DEV_TIMERS && tid == 0)
        for (int k = 0; k < 6; k++) atomicAdd(&g_deflate_t[k], dt[k]);

    WG_SYNC();
    return L.nout;

}

// ---- chunk bookkeeping ---------------------------------------------------------------

__device__ inline uint32_t chunks_of(uint64_t len) { return len ? (uint32_t)((len + kChunk - 1) / kChunk) : 1; }

// exclusive prefix of chunks per item (single workgroup)
__global__ __launch_bounds__(1024) void chunk_scan_kernel(const uint64_t *__restrict__ in_off, uint32_t n,
                                                           uint32_t *__restrict__ chunk_first)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    WG_SYNC();
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t c = i < n ? chunks_of(in_off[i + 1] - in_off[i]) : 0;
        uint32_t x = c;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        WG_SYNC();
        uint32_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) before += wsum[w];
        if (i < n) chunk_first[i] = before + x - c;
        WG_SYNC();
        if (threadIdx.x == 1023) carry = before + x;
        WG_SYNC();
    }
    if (threadIdx.x == 0) chunk_first[n] = carry;
}

// Which item a chunk belongs to, where it lies in the input, whether it ends the item's stream.
struct ChunkPlace {
    uint64_t in_at;    // offset of the chunk's first byte in `in`
    uint32_t len;      // its length
    bool ends_item;    // no chunk of the item follows
};

__device__ inline ChunkPlace place_chunk(const uint64_t *in_off, uint32_t n, const uint32_t *chunk_first, uint32_t ch)
{
    uint32_t lo = 0, hi = n;   // item with chunk_first[item] <= ch < chunk_first[item+1]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (chunk_first[mid] <= ch) lo = mid; else hi = mid;
    }
    const uint32_t item = lo, k = ch - chunk_first[item];
    const uint64_t ib = in_off[item], len = in_off[item + 1] - ib;
    const uint64_t cb = (uint64_t)k * kChunk;
    ChunkPlace c;
    c.in_at = ib + cb;
    c.len = (uint32_t)(len - cb < kChunk ? len - cb : kChunk);
    c.ends_item = cb + c.len >= len;
    return c;
}

// Match kernel.  Persistent workgroups that take the next chunk from a counter (*next_chunk, zero at the
// launch) whenever they are done with one: chunks differ in cost, and with a fixed stride a workgroup
// whose chunks are all of the slow kind (a batch of repeating buffers does that) holds the kernel up.  What
// a chunk compresses to does not depend on who takes it.  Per chunk the token units go to tokbuf +
// chunk * tok_stride, the block table to blk_tab / blk_hist + chunk * kMaxBlocks entries, the number of
// blocks to chunk_blocks.
template <bool kOptimal, bool kChains>
__global__ __launch_bounds__(kCT) void deflate_chunks_kernel(
    const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    const uint32_t *__restrict__ chunk_first, uint32_t chunk_slots, LevelParams lp,
    uint64_t scratch_stride, uint16_t *__restrict__ tokbuf, uint64_t tok_stride,
    uint2 *__restrict__ blk_tab, uint16_t *__restrict__ blk_hist, uint32_t *__restrict__ chunk_blocks,
    uint32_t *__restrict__ pairbuf, uint8_t *__restrict__ costbuf, uint32_t *__restrict__ next_chunk)
{
    __shared__ ChunkLds L;
    __shared__ uint32_t taken;
    uint32_t *mp = pairbuf ? pairbuf + (size_t)blockIdx.x * kChunk * kPairs : nullptr;
    uint8_t *blkcost = costbuf + (size_t)blockIdx.x * kMaxBlocks * kBlkCostStride;
    const uint32_t total_chunks = chunk_first[n];
    for (;;) {
        if (threadIdx.x == 0) taken = atomicAdd(next_chunk, 1u);
        WG_SYNC();
        const uint32_t ch = taken;
        WG_SYNC();   // (everyone has read it before the next turn overwrites it)
        if (ch >= total_chunks || ch >= chunk_slots) break;
        const ChunkPlace c = place_chunk(in_off, n, chunk_first, ch);
        // the caller understated the longest item: the chunk has no room in the token buffer or
        // in the chunk scratch (the emit kernel makes the item fall back or fail)
        if ((uint64_t)c.len + kScratchPad > scratch_stride) continue;
        const uint32_t nb = compress_chunk<kOptimal, kChains>(L, in + c.in_at, c.len, lp, tokbuf + (size_t)ch * tok_stride,
                                                              blk_tab + (size_t)ch * kMaxBlocks,
                                                              blk_hist + (size_t)ch * kMaxBlocks * kHistStride, mp, blkcost);
        if (threadIdx.x == 0) chunk_blocks[ch] = nb;
        WG_SYNC();
    }
}

// Emit kernel: one workgroup per chunk.  Per block: Huffman codes from the histogram, the block
// type decision, the header and the token bits (flush_block); level 0 and empty chunks are written
// here as well; then the sync flush marker between chunks (src/compress/mod.rs:662-681).  With
// little LDS and 512 lanes several chunks share a CU, so the serial parts of the code construction
// of one hide behind the others.
#ifndef RSX_EMIT_OCC
#define RSX_EMIT_OCC 8   // waves per SIMD the emit kernel is compiled for (four workgroups per CU)
#endif
__global__ __launch_bounds__(kET, RSX_EMIT_OCC) void deflate_emit_kernel(
    const uint8_t *__restrict__ in, const uint64_t *__restrict__ in_off, uint32_t n,
    const uint32_t *__restrict__ chunk_first, uint32_t chunk_slots, int stored_only,
    uint8_t *__restrict__ scratch, uint64_t scratch_stride, const uint16_t *__restrict__ tokbuf, uint64_t tok_stride,
    const uint2 *__restrict__ blk_tab, const uint16_t *__restrict__ blk_hist,
    const uint32_t *__restrict__ chunk_blocks, uint32_t *__restrict__ chunk_size, int finish, int static_code)
{
    __shared__ EmitLds L;
    const uint32_t tid = threadIdx.x;
    const uint32_t total_chunks = chunk_first[n];
    for (uint32_t ch = blockIdx.x; ch < total_chunks && ch < chunk_slots; ch += gridDim.x) {
        const ChunkPlace c = place_chunk(in_off, n, chunk_first, ch);
        if ((uint64_t)c.len + kScratchPad > scratch_stride) {
            // the caller understated the longest item: never write past the chunk's slot; the
            // size makes the item fall back to stored blocks or report InsufficientSpace (uniform)
            if (tid == 0) chunk_size[ch] = 0x7FFFFFFFu;
            continue;
        }
        // FlushMode::Sync leaves the item's stream open: every chunk ends byte aligned, none is final
        const bool last_chunk = finish && c.ends_item;
        const uint8_t *src = in + c.in_at;
        uint32_t *out_words = reinterpret_cast<uint32_t *>(scratch + (size_t)ch * scratch_stride);
        for (uint32_t i = tid; i < kStageWords; i += kET) L.stage[i] = 0;
        for (int i = tid; i < 260; i += kET) {
            uint32_t s = 0;
            while (s + 1 < 29 && k_len_base[s + 1] <= (uint32_t)i) s++;
            L.len_slot[i] = (uint8_t)s;
        }
        if (tid == 0) {
            L.out_bits = 0;
            L.stage_word0 = 0;
        }
        WG_SYNC();
        if (c.len == 0) {
            // an empty input is an empty final fixed-Huffman block: BFINAL=1, BTYPE=01, end-of-block (7 zero bits)
            if (tid == 0) {
                if (last_chunk) stage_or(L, 0, 3, 3);
                L.out_bits = last_chunk ? 10 : 0;
            }
            WG_SYNC();
        } else if (stored_only) {
            // level 0: stored blocks only (src/compress/mod.rs:1400-1449)
            for (uint32_t pos = 0; pos < c.len; pos += 65535) {
                const uint32_t blen = c.len - pos < 65535 ? c.len - pos : 65535;
                const uint64_t q0 = L.out_bits, q = (q0 + 3 + 7) & ~7ull;
                WG_SYNC();   // everyone has read out_bits
                if (tid == 0) {
                    stage_or(L, q0, (last_chunk && pos + blen >= c.len) ? 1 : 0, 3);
                    stage_or(L, q, blen | ((~blen & 0xFFFF) << 16), 32);
                    L.out_bits = q + 32;
                }
                WG_SYNC();
                stage_flush(L, out_words, q + 32);
                stage_bytes(L, out_words, src + pos, blen);
            }
        } else {
            const uint32_t nb = chunk_blocks[ch];
            const uint2 *tab = blk_tab + (size_t)ch * kMaxBlocks;
            const uint16_t *hist = blk_hist + (size_t)ch * kMaxBlocks * kHistStride;
            const uint16_t *tok = tokbuf + (size_t)ch * tok_stride;
            uint32_t start = 0, unit0 = 0;
            for (uint32_t k = 0; k < nb; k++) {
                const uint2 e = tab[k];   // (block end, token units)
                if (tid < 288) L.litlen_freq[tid] = hist[k * kHistStride + tid];
                else if (tid < 320) L.off_freq[tid - 288] = hist[k * kHistStride + tid];
                if (tid == 0) {
                    L.block_ntok = e.y;
                    L.block_start = start;
                }
                WG_SYNC();
                flush_block(L, src, e.x, last_chunk && k + 1 == nb, tok + unit0, out_words, static_code != 0);
                start = e.x;
                unit0 += e.y;
            }
        }
        // chunk trailer: sync flush between chunks (src/compress/mod.rs:662-681)
        {
            const uint64_t p0 = L.out_bits;
            const uint64_t p = last_chunk ? (p0 + 7) & ~7ull : ((p0 + 3 + 7) & ~7ull) + 32;
            WG_SYNC();   // everyone has read out_bits
            if (tid == 0) {
                if (!last_chunk) stage_or(L, p - 32, 0xFFFF0000u, 32);   // empty stored block header, byte aligned
                L.out_bits = p;
            }
            WG_SYNC();
            stage_flush(L, out_words, p);
            // the unit that is still open (zero bits behind the last byte)
            if (tid == 0) {
                reinterpret_cast<uint4 *>(out_words + L.stage_word0)[0] = reinterpret_cast<uint4 *>(L.stage)[0];
                chunk_size[ch] = (uint32_t)(p >> 3);
            }
        }
        WG_SYNC();
    }
}

// ---- gather: the item's chunks back to back behind the container header, then the trailer ----
// deflate_plan_kernel (one workgroup per item) sizes the item, chooses between its compressed
// chunks and plain stored blocks, writes header, trailer, size and status and gives every chunk
// its place in the output; deflate_place_kernel (one workgroup per chunk) moves the bytes, so one
// large item is copied by as many workgroups as it has chunks.

constexpr uint32_t kModeChunks = 0, kModeStored = 1, kModeFailed = 2;

__global__ __launch_bounds__(256) void deflate_plan_kernel(
    int format, int level, const uint64_t *__restrict__ in_off, uint32_t n,
    const uint32_t *__restrict__ chunk_first, const uint32_t *__restrict__ chunk_size,
    const uint32_t *__restrict__ sums, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    uint64_t *__restrict__ out_sizes, int32_t *__restrict__ status, uint64_t *__restrict__ chunk_dst,
    uint32_t *__restrict__ item_mode, int finish)
{
    __shared__ uint64_t wsum[4];
    __shared__ uint64_t carry;
    const uint32_t item = blockIdx.x, tid = threadIdx.x;
    if (item >= n) return;
    const uint32_t c0 = chunk_first[item], c1 = chunk_first[item + 1];
    const uint64_t slot = out_off[item + 1] - out_off[item];
    uint8_t *o = out + out_off[item];
    const uint32_t hdr = format == RSX_FORMAT_ZLIB ? 2 : format == RSX_FORMAT_GZIP ? 10 : 0;
    const uint32_t trl = format == RSX_FORMAT_ZLIB ? 4 : format == RSX_FORMAT_GZIP ? 8 : 0;
    // exclusive prefix of the chunk sizes -> place of every chunk behind the header
    if (tid == 0) carry = hdr;
    __syncthreads();
    for (uint32_t base = c0; base < c1; base += 256) {
        const uint32_t c = base + tid;
        const uint64_t v = c < c1 ? chunk_size[c] : 0;
        uint64_t x = v;
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t y = __shfl_up(x, d);
            if ((int)(tid & 63) >= d) x += y;
        }
        if ((tid & 63) == 63) wsum[tid >> 6] = x;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < (tid >> 6); w++) before += wsum[w];
        if (c < c1) chunk_dst[c] = before + x - v;
        __syncthreads();
        if (tid == 255) carry = before + x;
        __syncthreads();
    }
    uint64_t total = carry + trl;
    // The whole item as stored blocks of 65535 bytes (the level-0 format,
    // src/compress/mod.rs:1400-1464) when that is smaller: incompressible items then fit the
    // compress bound however many chunks they have (a chunk of stored blocks pays for a short
    // fifth block and its sync flush).  An open stream still ends with the empty stored block.
    const uint64_t ilen = in_off[item + 1] - in_off[item];
    const uint64_t nblk = ilen ? (ilen + 65534) / 65535 : 1;
    const uint64_t stored_total = hdr + trl + ilen + 5 * nblk + (finish ? 0 : 5);
    const bool stored = stored_total <= slot && (total > slot || stored_total < total);
    if (total > slot && !stored) {
        // CompressResult::InsufficientSpace (src/compress/mod.rs:756-758)
        if (tid == 0) {
            status[item] = RSX_COMPRESS_INSUFFICIENT_SPACE;
            out_sizes[item] = 0;
            item_mode[item] = kModeFailed;
        }
        return;
    }
    if (stored) total = stored_total;
    if (tid == 0) {
        item_mode[item] = stored ? kModeStored : kModeChunks;
        uint64_t at = total - trl;
        if (stored && !finish) {
            uint8_t *q = o + at - 5;
            q[0] = 0; q[1] = 0; q[2] = 0; q[3] = 0xFF; q[4] = 0xFF;
        }
        if (format == RSX_FORMAT_ZLIB) {
            // src/compress/mod.rs:2257-2269, :2288-2296
            uint32_t h = (8u << 8) | (7u << 12);
            uint32_t hint = level < 2 ? 0 : level < 6 ? 1 : level < 8 ? 2 : 3;
            h |= hint << 6;
            h |= 31 - (h % 31);
            o[0] = (uint8_t)(h >> 8);
            o[1] = (uint8_t)h;
            uint32_t a = sums[item];
            o[at] = (uint8_t)(a >> 24);
            o[at + 1] = (uint8_t)(a >> 16);
            o[at + 2] = (uint8_t)(a >> 8);
            o[at + 3] = (uint8_t)a;
        } else if (format == RSX_FORMAT_GZIP) {
            // src/compress/mod.rs:2309-2327, :2339-2355
            o[0] = 0x1F; o[1] = 0x8B; o[2] = 8; o[3] = 0;
            o[4] = o[5] = o[6] = o[7] = 0;
            o[8] = level < 2 ? 4 : level >= 8 ? 2 : 0;
            o[9] = 255;
            uint32_t c = sums[item], isz = (uint32_t)ilen;
            for (int k = 0; k < 4; k++) {
                o[at + k] = (uint8_t)(c >> (8 * k));
                o[at + 4 + k] = (uint8_t)(isz >> (8 * k));
            }
        }
        status[item] = 0;
        out_sizes[item] = total;
    }
}

// All lanes of a workgroup: count bytes from src to dst, any alignment on both sides: the
// destination is written as aligned dwords made of two aligned source dwords.
__device__ inline void copy_bytes(uint8_t *dst, const uint8_t *src, uint64_t count)
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

__global__ __launch_bounds__(256) void deflate_place_kernel(
    const uint64_t *__restrict__ in_off, uint32_t n, const uint32_t *__restrict__ chunk_first,
    uint32_t chunk_slots, const uint8_t *__restrict__ scratch, uint64_t scratch_stride,
    const uint32_t *__restrict__ chunk_size, const uint64_t *__restrict__ chunk_dst,
    const uint32_t *__restrict__ item_mode, uint8_t *__restrict__ out, const uint64_t *__restrict__ out_off,
    const uint8_t *__restrict__ in, int format, int finish)
{
    const uint32_t total_chunks = chunk_first[n];
    for (uint32_t ch = blockIdx.x; ch < total_chunks && ch < chunk_slots; ch += gridDim.x) {
        uint32_t lo = 0, hi = n;   // item with chunk_first[item] <= ch < chunk_first[item+1]
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (chunk_first[mid] <= ch) lo = mid; else hi = mid;
        }
        const uint32_t item = lo, mode = item_mode[item];
        uint8_t *o = out + out_off[item];
        if (mode == kModeChunks) {
            copy_bytes(o + chunk_dst[ch], scratch + (size_t)ch * scratch_stride, chunk_size[ch]);
        } else if (mode == kModeStored) {
            // stored blocks k, k + chunks, ... of the item
            const uint32_t hdr = format == RSX_FORMAT_ZLIB ? 2 : format == RSX_FORMAT_GZIP ? 10 : 0;
            const uint32_t k = ch - chunk_first[item], step = chunk_first[item + 1] - chunk_first[item];
            const uint64_t ilen = in_off[item + 1] - in_off[item];
            const uint64_t nblk = ilen ? (ilen + 65534) / 65535 : 1;
            const uint8_t *src = in + in_off[item];
            for (uint64_t b = k; b < nblk; b += step) {
                const uint64_t pos = b * 65535;
                const uint32_t blen = (uint32_t)(ilen - pos < 65535 ? ilen - pos : 65535);
                uint8_t *q = o + hdr + b * (65535 + 5);
                if (threadIdx.x == 0) {
                    q[0] = (finish && b == nblk - 1) ? 1 : 0;
                    q[1] = (uint8_t)blen;
                    q[2] = (uint8_t)(blen >> 8);
                    q[3] = (uint8_t)~blen;
                    q[4] = (uint8_t)(~blen >> 8);
                }
                copy_bytes(q + 5, src + pos, blen);
            }
        }
    }
}

uint64_t chunk_upper_bound(size_t n, uint64_t total_in) { return total_in / kChunk + n; }

}  // namespace

// upper bound on the workgroups deflate_launch starts
static uint32_t deflate_grid_bound(uint64_t chunks) { return (uint32_t)(chunks < 1024 ? chunks : 1024); }

// token units per chunk in the token buffer: at most one per byte, and the last 16-byte piece of a
// batch may end up to 7 units behind them.  Rounded like the chunk scratch below: a chunk that passes the
// kernels' guard against an understated max_item (its length fits the scratch stride) then fits here too.
static uint64_t tok_stride_units(uint64_t max_item)
{
    const uint64_t longest = max_item < kChunk ? max_item : kChunk;
    return align_up(longest, 256) + 64;
}

// bytes of scratch per chunk: no chunk is longer than the longest item
static uint64_t scratch_stride(uint64_t max_item)
{
    const uint64_t longest = max_item < kChunk ? max_item : kChunk;
    return align_up(longest, 256) + kScratchPad;
}

size_t deflate_ws_bytes(size_t n, uint64_t total_in, uint64_t max_item, int level)
{
    const uint64_t chunks = chunk_upper_bound(n, total_in);
    const size_t pairs = level >= 10 ? align_up((size_t)deflate_grid_bound(chunks) * kChunk * kPairs * 4, 256) : 0;
    const uint64_t stride = scratch_stride(max_item);
    return align_up(chunks * stride, 256) + align_up((chunks + 1) * 4, 256) * 2 +
           align_up((n + 1) * 4, 256) + align_up(chunks * tok_stride_units(max_item) * 2, 256) + align_up(chunks * kMaxBlocks * sizeof(uint2), 256) +
           align_up(chunks * kMaxBlocks * kHistStride * 2, 256) + align_up((chunks + 1) * 4, 256) +
           checksum_ws_bytes(n, total_in) + pairs +
           align_up((size_t)1024 * kMaxBlocks * kBlkCostStride, 256) + align_up((chunks + 1) * 8, 256) +
           align_up((n + 1) * 4, 256) + 256 + 4096;
}

int deflate_launch(rsx_engine *e, int format, int level, const uint8_t *in, const uint64_t *in_off,
                   size_t n, uint64_t total_in, uint64_t max_item, uint8_t *out,
                   const uint64_t *out_off, uint64_t *out_sizes, int32_t *status, int flush,
                   bool trusted_max)
{
    if (n == 0) return RSX_OK;
    if (n > 0x7FFFFFFFull || max_item >= (1ull << 32)) {
        e->err = "batch too large (items must be < 4 GiB, n < 2^31)";
        return RSX_E_INVALID;
    }
    const int num_cus = e->num_cus > 0 ? e->num_cus : 256;
    const uint64_t chunks = chunk_upper_bound(n, total_in);
    const uint64_t stride = scratch_stride(max_item);
    uint32_t grid = (uint32_t)(chunks < (uint64_t)num_cus ? chunks : (uint64_t)num_cus);
    if (grid > 1024) grid = 1024;
    uint8_t *scratch = static_cast<uint8_t *>(ws_take(e, chunks * stride));
    uint32_t *chunk_size = static_cast<uint32_t *>(ws_take(e, (chunks + 1) * 4));
    uint32_t *chunk_first = static_cast<uint32_t *>(ws_take(e, (n + 1) * 4));
    const uint64_t tok_units = tok_stride_units(max_item);
    uint16_t *tokbuf = static_cast<uint16_t *>(ws_take(e, chunks * tok_units * 2));
    uint2 *blk_tab = static_cast<uint2 *>(ws_take(e, chunks * kMaxBlocks * sizeof(uint2)));
    uint16_t *blk_hist = static_cast<uint16_t *>(ws_take(e, chunks * kMaxBlocks * kHistStride * 2));
    uint32_t *chunk_blocks = static_cast<uint32_t *>(ws_take(e, (chunks + 1) * 4));
    uint32_t *sums = static_cast<uint32_t *>(ws_take(e, (n + 1) * 4));
    uint8_t *costbuf = static_cast<uint8_t *>(ws_take(e, (size_t)grid * kMaxBlocks * kBlkCostStride));
    uint64_t *chunk_dst = static_cast<uint64_t *>(ws_take(e, (chunks + 1) * 8));
    uint32_t *item_mode = static_cast<uint32_t *>(ws_take(e, (n + 1) * 4));
    uint32_t *next_chunk = static_cast<uint32_t *>(ws_take(e, 256));
    uint32_t *pairbuf = nullptr;
    if (level >= 10) {
        pairbuf = static_cast<uint32_t *>(ws_take(e, (size_t)grid * kChunk * kPairs * 4));
        if (!pairbuf) {
            e->err = "workspace exhausted";
            return RSX_E_NOMEM;
        }
    }
    if (!scratch || !chunk_size || !chunk_first || !tokbuf || !blk_tab || !blk_hist || !chunk_blocks || !sums ||
        !costbuf || !chunk_dst || !item_mode || !next_chunk) {
        e->err = "workspace exhausted";
        return RSX_E_NOMEM;
    }
    LevelParams lp = level_params(level);
    chunk_scan_kernel<<<1, 1024, 0, e->stream>>>(in_off, (uint32_t)n, chunk_first);
    RSX_HIP(e, hipMemsetAsync(next_chunk, 0, sizeof(uint32_t), e->stream));
    (void)hipEventRecord(e->ev_begin, e->stream);
#define RSX_LAUNCH_CHUNKS(OPT, CH)                                                                          \
    deflate_chunks_kernel<OPT, CH><<<grid, kCT, 0, e->stream>>>(in, in_off, (uint32_t)n, chunk_first,       \
                                                                (uint32_t)chunks, lp, stride, tokbuf, tok_units, \
                                                                blk_tab, blk_hist, chunk_blocks, pairbuf, costbuf, \
                                                                next_chunk)
    if (level == 0) { /* stored blocks only: nothing to match */ }
    else if (lp.optimal) RSX_LAUNCH_CHUNKS(true, true);
    else if (lp.chains) RSX_LAUNCH_CHUNKS(false, true);
    else RSX_LAUNCH_CHUNKS(false, false);
#undef RSX_LAUNCH_CHUNKS
    {
        const uint32_t emit_grid = (uint32_t)(chunks < 65536 ? chunks : 65536);
        deflate_emit_kernel<<<emit_grid, kET, 0, e->stream>>>(in, in_off, (uint32_t)n, chunk_first, (uint32_t)chunks,
                                                               level == 0, scratch, stride, tokbuf, tok_units, blk_tab,
                                                               blk_hist, chunk_blocks, chunk_size,
                                                               flush == RSX_FLUSH_FINISH, level == 1);
    }
    e->timed = true;
    RSX_HIP(e, hipGetLastError());
    if (RSX_DEV_TIMERS && getenv("RSX_DEFLATE_STATS")) {
        unsigned long long t[16];
        RSX_HIP(e, hipStreamSynchronize(e->stream));
        RSX_HIP(e, hipMemcpyFromSymbol(t, HIP_SYMBOL(g_deflate_t), sizeof t));
        fprintf(stderr, "[deflate] cycles/chunk: slide %.0f insert %.0f search %.0f parse %.0f emit %.0f flush %.0f (codes %.0f header %.0f); "
                        "wave 0: near %.0f rows %.0f cycles, %.1f near rounds and %.1f row rounds per batch, per lane %.2f candidates, %.2f compared in full; %.2f blocks per chunk, %.3f of them stored\n",
                t[0] / (double)chunks, t[1] / (double)chunks, t[2] / (double)chunks, t[3] / (double)chunks,
                t[4] / (double)chunks, t[5] / (double)chunks, t[6] / (double)chunks, t[7] / (double)chunks,
                t[8] / (double)chunks, t[9] / (double)chunks, t[10] / (double)chunks / 256, t[11] / (double)chunks / 256,
                t[12] / (double)chunks / 256 / 64, t[13] / (double)chunks / 256 / 64, t[14] / (double)chunks,
                t[15] / (double)(t[14] ? t[14] : 1));
        unsigned long long z[16] = {0};
        RSX_HIP(e, hipMemcpyToSymbol(HIP_SYMBOL(g_deflate_t), z, sizeof z));
    }
    if (format != RSX_FORMAT_DEFLATE) {
        int kind = format == RSX_FORMAT_GZIP ? RSX_CHECKSUM_CRC32 : RSX_CHECKSUM_ADLER32;
        int r = checksum_launch(e, kind, format == RSX_FORMAT_GZIP ? 0u : 1u, in, in_off, nullptr, n,
                                total_in, max_item, sums, trusted_max);
        if (r != RSX_OK) return r;
    }
    deflate_plan_kernel<<<(uint32_t)n, 256, 0, e->stream>>>(format, level, in_off, (uint32_t)n, chunk_first,
                                                            chunk_size, sums, out, out_off, out_sizes, status,
                                                            chunk_dst, item_mode, flush == RSX_FLUSH_FINISH);
    const uint32_t place_grid = (uint32_t)(chunks < 65536 ? chunks : 65536);
    deflate_place_kernel<<<place_grid, 256, 0, e->stream>>>(in_off, (uint32_t)n, chunk_first, (uint32_t)chunks,
                                                            scratch, stride, chunk_size, chunk_dst, item_mode,
                                                            out, out_off, in, format, flush == RSX_FLUSH_FINISH);
    (void)hipEventRecord(e->ev_end, e->stream);   // (the timed span is the whole call: match, emit, checksum, plan, place)
    RSX_HIP(e, hipGetLastError());
    return RSX_OK;
}
