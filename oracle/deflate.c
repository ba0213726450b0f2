/*
 * deflate.c -- raw DEFLATE / zlib / gzip compression restated on the CPU for
 * levels 0-12 (oracle; test infrastructure only, see rsx_oracle.h).
 *
 * Follows src/compress/mod.rs (driver, greedy/lazy parse, near-optimal parse of
 * levels 10-12, block splitting, dynamic header, sequence writer),
 * src/compress/matchfinder.rs (hash-chain, single-probe hash-table and
 * binary-tree match finders) and src/compress/huffman_comp.rs
 * (length-limited canonical Huffman code).  Like the reference it has no
 * stored-block fallback above level 0: a stream that does not fit the output
 * buffer reports InsufficientSpace (src/compress/mod.rs:641-644).
 */
#include "rsx_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MIN_MATCH 3
#define MAX_MATCH 258
#define MAX_OFFSET 32768
#define HASH_ORDER 15
#define HASH_SIZE (1u << HASH_ORDER)
#define WINDOW_SIZE 32768u
#define NUM_LITLEN 288
#define NUM_OFFSET 32
#define MAX_LITLEN_CODEWORD_LEN 14 /* src/compress/mod.rs:127 */
#define MAX_OFFSET_CODEWORD_LEN 15
#define MAX_PRE_CODEWORD_LEN 7
#define MIN_BLOCK_LENGTH 5000     /* src/common.rs:68 */
#define SOFT_MAX_BLOCK_LENGTH 300000
#define CHUNK_SIZE (256u * 1024u) /* src/compress/mod.rs:699-700 */

/* ---- bit writer (src/compress/bitstream.rs) ----------------------------- */

typedef struct {
    uint8_t *out;
    size_t cap, idx;
    uint64_t buf;
    unsigned cnt;
    int overflow;
} bitw;

static inline void bw_put(bitw *w, uint32_t v, unsigned n)
{
    w->buf |= (uint64_t)v << w->cnt;
    w->cnt += n;
    if (w->cnt >= 32) {
        if (w->idx + 4 <= w->cap) {
            uint32_t lo = (uint32_t)w->buf;
            memcpy(w->out + w->idx, &lo, 4);
            w->idx += 4;
        } else {
            w->overflow = 1;
        }
        w->buf >>= 32;
        w->cnt -= 32;
    }
}

/* bitstream.rs:194-222: write out whole and partial bytes, byte-align */
static inline void bw_flush(bitw *w)
{
    while (w->cnt > 0) {
        if (w->idx < w->cap)
            w->out[w->idx++] = (uint8_t)w->buf;
        else
            w->overflow = 1;
        w->buf >>= 8;
        w->cnt = w->cnt > 8 ? w->cnt - 8 : 0;
    }
    w->buf = 0;
}

/* ---- length / offset slot tables (src/compress/mod.rs:16-105) ----------- */

static const uint16_t length_base[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,
                                         15, 17, 19, 23, 27, 31, 35, 43, 51,  59,
                                         67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t length_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2,
                                         2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t offset_base[30] = {1,    2,    3,    4,    5,    7,    9,    13,   17,   25,
                                         33,   49,   65,   97,   129,  193,  257,  385,  513,  769,
                                         1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385,
                                         24577};
static const uint8_t offset_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3,  3,  4,  4,  5,  5,  6,
                                         6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint8_t length_slot_tab[MAX_MATCH + 1];
static uint8_t offset_slot_tab[512];
static int tabs_ready;

static void init_tabs(void)
{
    for (int s = 0; s < 29; s++) {
        int hi = (s == 28) ? 258 : length_base[s] + (1 << length_extra[s]) - 1;
        /* slot 27 covers 227..257; 258 has its own slot 28 */
        if (s == 27) hi = 257;
        for (int l = length_base[s]; l <= hi; l++)
            length_slot_tab[l] = (uint8_t)s;
    }
    /* src/compress/mod.rs:69-105 */
    for (unsigned off = 1; off <= 256; off++) {
        unsigned slot = 0;
        while (slot + 1 < 30 && offset_base[slot + 1] <= off) slot++;
        offset_slot_tab[off] = (uint8_t)slot;
    }
    for (unsigned k = 2; k < 256; k++) {
        unsigned off = (k << 7) + 1;
        unsigned slot = 0;
        while (slot + 1 < 30 && offset_base[slot + 1] <= off) slot++;
        offset_slot_tab[256 + k] = (uint8_t)slot;
    }
    __atomic_store_n(&tabs_ready, 1, __ATOMIC_RELEASE);
}

/* src/compress/mod.rs:2196-2207 */
static inline unsigned offset_slot(unsigned offset)
{
    return offset <= 256 ? offset_slot_tab[offset] : offset_slot_tab[256 + ((offset - 1) >> 7)];
}

/* ---- Huffman code construction (src/compress/huffman_comp.rs) ----------- */

#define NUM_SYMBOL_BITS 10
#define SYMBOL_MASK ((1u << NUM_SYMBOL_BITS) - 1)
#define FREQ_MASK (~SYMBOL_MASK)

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* huffman_comp.rs:8-36 */
static unsigned sort_symbols(unsigned num_syms, const uint32_t *freqs, uint8_t *lens,
                             uint32_t *symout)
{
    unsigned counters[NUM_LITLEN] = {0};
    for (unsigned sym = 0; sym < num_syms; sym++) {
        unsigned f = freqs[sym] < num_syms - 1 ? freqs[sym] : num_syms - 1;
        counters[f]++;
    }
    unsigned pos = 0;
    for (unsigned i = 1; i < num_syms; i++) {
        unsigned c = counters[i];
        counters[i] = pos;
        pos += c;
    }
    unsigned num_used = pos;
    for (unsigned sym = 0; sym < num_syms; sym++) {
        uint32_t f = freqs[sym];
        if (f != 0) {
            unsigned b = f < num_syms - 1 ? f : num_syms - 1;
            symout[counters[b]++] = sym | (f << NUM_SYMBOL_BITS);
        } else {
            lens[sym] = 0;
        }
    }
    unsigned hs = counters[num_syms - 2], he = counters[num_syms - 1];
    if (he > hs)
        qsort(symout + hs, he - hs, sizeof(uint32_t), cmp_u32);
    return num_used;
}

/* huffman_comp.rs:38-62 */
static void build_tree(uint32_t *a, unsigned sym_count)
{
    unsigned last = sym_count - 1, i = 0, b = 0, e = 0;
    while (e < last) {
        uint32_t nf;
        if (i < last && (b == e || (a[i + 1] & FREQ_MASK) <= (a[b] & FREQ_MASK))) {
            nf = (a[i] & FREQ_MASK) + (a[i + 1] & FREQ_MASK);
            i += 2;
        } else if (b + 2 <= e && (i > last || (a[b + 1] & FREQ_MASK) < (a[i] & FREQ_MASK))) {
            nf = (a[b] & FREQ_MASK) + (a[b + 1] & FREQ_MASK);
            a[b] = (e << NUM_SYMBOL_BITS) | (a[b] & SYMBOL_MASK);
            a[b + 1] = (e << NUM_SYMBOL_BITS) | (a[b + 1] & SYMBOL_MASK);
            b += 2;
        } else {
            nf = (a[i] & FREQ_MASK) + (a[b] & FREQ_MASK);
            a[b] = (e << NUM_SYMBOL_BITS) | (a[b] & SYMBOL_MASK);
            i += 1;
            b += 1;
        }
        a[e] = nf | (a[e] & SYMBOL_MASK);
        e++;
    }
}

/* huffman_comp.rs:64-89 */
static void compute_length_counts(uint32_t *a, unsigned root_idx, unsigned *len_counts,
                                  unsigned max_len)
{
    for (unsigned l = 0; l <= max_len; l++) len_counts[l] = 0;
    len_counts[1] = 2;
    a[root_idx] &= SYMBOL_MASK;
    for (int node = (int)root_idx - 1; node >= 0; node--) {
        unsigned parent = a[node] >> NUM_SYMBOL_BITS;
        unsigned depth = (a[parent] >> NUM_SYMBOL_BITS) + 1;
        a[node] = (a[node] & SYMBOL_MASK) | (depth << NUM_SYMBOL_BITS);
        if (depth >= max_len) {
            depth = max_len - 1;
            while (len_counts[depth] == 0) depth--;
        }
        len_counts[depth]--;
        len_counts[depth + 1] += 2;
    }
}

static uint32_t reverse_codeword(uint32_t c, unsigned len)
{
    uint32_t r = 0;
    for (unsigned i = 0; i < len; i++) r |= ((c >> i) & 1) << (len - 1 - i);
    return r;
}

/* huffman_comp.rs:95-123 */
static void gen_codewords(uint32_t *a, uint8_t *lens, const unsigned *len_counts,
                          unsigned max_len, unsigned num_syms)
{
    unsigned next[16] = {0};
    unsigned i = 0;
    for (unsigned len = max_len; len >= 1; len--)
        for (unsigned c = len_counts[len]; c > 0; c--)
            lens[a[i++] & SYMBOL_MASK] = (uint8_t)len;
    for (unsigned len = 2; len <= max_len; len++)
        next[len] = (next[len - 1] + len_counts[len - 1]) << 1;
    for (unsigned sym = 0; sym < num_syms; sym++)
        if (lens[sym])
            a[sym] = reverse_codeword(next[lens[sym]]++, lens[sym]);
}

/* huffman_comp.rs:125-155 */
static void make_huffman_code(unsigned num_syms, unsigned max_len, const uint32_t *freqs,
                              uint8_t *lens, uint32_t *codewords)
{
    unsigned used = sort_symbols(num_syms, freqs, lens, codewords);
    if (used < 2) {
        unsigned sym = used ? (codewords[0] & SYMBOL_MASK) : 0;
        unsigned nz = sym ? sym : 1;
        codewords[0] = 0;
        lens[0] = 1;
        codewords[nz] = 1;
        lens[nz] = 1;
        return;
    }
    build_tree(codewords, used);
    unsigned len_counts[16];
    compute_length_counts(codewords, used - 2, len_counts, max_len);
    gen_codewords(codewords, lens, len_counts, max_len, num_syms);
}

/* ---- block split statistics (src/compress/mod.rs:271-416) --------------- */

#define NUM_OBS_TYPES 14
typedef struct {
    uint32_t new_obs[NUM_OBS_TYPES], obs[NUM_OBS_TYPES];
    uint32_t num_new, num;
} split_stats;

static const uint8_t slot_to_obs[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                        1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 0, 0};

static inline unsigned bsr32(uint32_t v) { return 31 - (unsigned)__builtin_clz(v); }

static inline void observe_literal(split_stats *s, uint8_t lit)
{
    s->new_obs[lit >> 5]++;
    s->num_new++;
}

static inline void observe_match(split_stats *s, unsigned len, unsigned slot)
{
    s->new_obs[8 + (len >= 8)]++;
    s->new_obs[10 + slot_to_obs[slot]]++;
    s->num_new += 2;
}

static int do_end_block_check(const split_stats *s, size_t block_len)
{
    if (s->num == 0) return 0;
    uint32_t old_bits = 0, new_bits = 0;
    unsigned l_obs = bsr32(s->num), l_new = bsr32(s->num_new);
    for (int i = 0; i < NUM_OBS_TYPES; i++) {
        uint32_t n = s->new_obs[i];
        if (n) {
            unsigned lo = bsr32(s->obs[i] + 1), ln = bsr32(n + 1);
            old_bits += n * (l_obs > lo ? l_obs - lo : 0);
            new_bits += n * (l_new > ln ? l_new - ln : 0);
        }
    }
    return ((int32_t)old_bits - (int32_t)new_bits) > (int32_t)block_len / 16;
}

static int should_end_block(split_stats *s, size_t block_len, size_t remaining)
{
    if (s->num_new < 2048 && block_len < SOFT_MAX_BLOCK_LENGTH) return 0;
    if (remaining <= MIN_BLOCK_LENGTH) return 0;
    if (block_len >= SOFT_MAX_BLOCK_LENGTH) return 1;
    if (block_len >= MIN_BLOCK_LENGTH) {
        if (do_end_block_check(s, block_len)) return 1;
        for (int i = 0; i < NUM_OBS_TYPES; i++) {
            s->obs[i] += s->new_obs[i];
            s->new_obs[i] = 0;
        }
        s->num += s->num_new;
        s->num_new = 0;
    }
    return 0;
}

/* ---- compressor state ---------------------------------------------------- */

#define BT_HASH_SIZE (1u << 16)   /* matchfinder.rs:1319-1320 */
#define DP_CAP 300000u            /* hard cap of a block, mod.rs:278-416 */
#define BT_MAX_MATCHES 512u       /* a position reports at most one match per length */

typedef struct {
    uint32_t litrunlen;
    uint16_t length; /* len | offset_slot << 9 */
    uint16_t offset;
} sequence;

struct orc_compressor {
    int level;
    unsigned max_depth, nice_len;
    int32_t *hash_tab;
    uint16_t *prev_tab;
    size_t base_offset;
    uint32_t litlen_freqs[NUM_LITLEN], offset_freqs[NUM_OFFSET];
    uint8_t litlen_lens[NUM_LITLEN], offset_lens[NUM_OFFSET];
    uint32_t litlen_codes[NUM_LITLEN], offset_codes[NUM_OFFSET];
    sequence *seqs;
    size_t nseq, seq_cap;
    split_stats stats;
    /* levels 10-12 (src/compress/mod.rs:478-503) */
    int32_t (*bt_hash3)[2];
    int32_t *bt_hash4;
    int32_t (*bt_child)[2];
    uint32_t *dp_costs, *dp_path;
    uint16_t (*matches)[2];
    uint32_t length_costs[MAX_MATCH + 1], offset_slot_costs[NUM_OFFSET];
};

/* src/compress/mod.rs:543-602 */
static void init_params(orc_compressor *c)
{
    static const unsigned depth[13] = {0, 2, 6, 12, 16, 16, 35, 100, 300, 600, 35, 100, 300};
    static const unsigned nice[13] = {0, 32, 10, 14, 30, 30, 65, 130, 258, 258, 75, 150, 258};
    int l = c->level > 12 ? 12 : c->level;
    c->max_depth = depth[l];
    c->nice_len = nice[l];
}

orc_compressor *orc_compressor_new(int level)
{
    if (level < 0 || level > 12) return NULL;
    if (!__atomic_load_n(&tabs_ready, __ATOMIC_ACQUIRE)) init_tabs();
    orc_compressor *c = calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->level = level;
    init_params(c);
    c->seq_cap = 50000; /* src/compress/mod.rs:486 */
    c->seqs = malloc(c->seq_cap * sizeof(sequence));
    if (level >= 10) {
        /* matchfinder.rs:1317-1325, mod.rs:488-503 */
        c->bt_hash3 = malloc(BT_HASH_SIZE * sizeof(*c->bt_hash3));
        c->bt_hash4 = malloc(BT_HASH_SIZE * sizeof(int32_t));
        c->bt_child = calloc(WINDOW_SIZE, sizeof(*c->bt_child));
        c->dp_costs = malloc((DP_CAP + 1) * sizeof(uint32_t));
        c->dp_path = malloc((DP_CAP + 1) * sizeof(uint32_t));
        c->matches = malloc(BT_MAX_MATCHES * sizeof(*c->matches));
        if (!c->bt_hash3 || !c->bt_hash4 || !c->bt_child || !c->dp_costs || !c->dp_path ||
            !c->matches || !c->seqs) {
            orc_compressor_free(c);
            return NULL;
        }
        memset(c->bt_hash3, 0xFF, BT_HASH_SIZE * sizeof(*c->bt_hash3));
        memset(c->bt_hash4, 0xFF, BT_HASH_SIZE * sizeof(int32_t));
        return c;
    }
    c->hash_tab = malloc(HASH_SIZE * sizeof(int32_t));
    c->prev_tab = calloc(WINDOW_SIZE, sizeof(uint16_t));
    if (!c->hash_tab || !c->prev_tab || !c->seqs) {
        orc_compressor_free(c);
        return NULL;
    }
    memset(c->hash_tab, 0xFF, HASH_SIZE * sizeof(int32_t));
    return c;
}

void orc_compressor_free(orc_compressor *c)
{
    if (!c) return;
    free(c->hash_tab);
    free(c->prev_tab);
    free(c->seqs);
    free(c->bt_hash3);
    free(c->bt_hash4);
    free(c->bt_child);
    free(c->dp_costs);
    free(c->dp_path);
    free(c->matches);
    free(c);
}

static void push_seq(orc_compressor *c, uint32_t litrunlen, unsigned len, unsigned offset,
                     unsigned slot)
{
    if (c->nseq == c->seq_cap) {
        c->seq_cap *= 2;
        c->seqs = realloc(c->seqs, c->seq_cap * sizeof(sequence));
    }
    sequence *s = &c->seqs[c->nseq++];
    s->litrunlen = litrunlen;
    s->length = (uint16_t)(len | (slot << 9));
    s->offset = (uint16_t)offset;
}

/* ---- match finders (src/compress/matchfinder.rs) ------------------------- */

static inline uint32_t load24(const uint8_t *p)
{
    return p[0] | (p[1] << 8) | ((uint32_t)p[2] << 16);
}

static inline uint32_t hash3(uint32_t v24) { return (v24 * 0x1E35A7BDu) >> (32 - HASH_ORDER); }

static inline unsigned match_len(const uint8_t *a, const uint8_t *b, unsigned max_len)
{
    unsigned n = 0;
    while (n + 8 <= max_len) {
        uint64_t x, y;
        memcpy(&x, a + n, 8);
        memcpy(&y, b + n, 8);
        if (x != y) return n + (unsigned)(__builtin_ctzll(x ^ y) >> 3);
        n += 8;
    }
    while (n < max_len && a[n] == b[n]) n++;
    return n;
}

/* insert `pos` into the chain and return the previous head (or -1);
 * matchfinder.rs:784-805 and :1020-1057 */
static inline int32_t hc_insert(orc_compressor *c, const uint8_t *data, size_t pos)
{
    uint32_t h = hash3(load24(data + pos));
    size_t abs_pos = c->base_offset + pos;
    int32_t cur = c->hash_tab[h];
    c->hash_tab[h] = (int32_t)abs_pos;
    if (cur == -1 || (size_t)cur < c->base_offset) {
        c->prev_tab[abs_pos & (WINDOW_SIZE - 1)] = 0;
        return -1;
    }
    size_t d = abs_pos - (size_t)cur;
    c->prev_tab[abs_pos & (WINDOW_SIZE - 1)] = d > 0xFFFF ? 0 : (uint16_t)d;
    return cur;
}

/* matchfinder.rs:754-891 */
static unsigned hc_find_match(orc_compressor *c, const uint8_t *data, size_t len, size_t pos,
                              unsigned *offset_ret)
{
    *offset_ret = 0;
    if (pos + 3 > len) return 0;
    int safe4 = pos + 4 <= len;
    const uint8_t *src = data + pos;
    uint32_t src4 = 0, src3;
    if (safe4) {
        memcpy(&src4, src, 4);
        src3 = src4 & 0xFFFFFF;
    } else {
        src3 = load24(src);
    }
    size_t abs_pos = c->base_offset + pos;
    int32_t cur = hc_insert(c, data, pos);
    if (cur == -1) return 0;

    unsigned best_len = 0, best_off = 0, depth = 0;
    while (cur != -1 && depth < c->max_depth) {
        size_t p_abs = (size_t)cur;
        if (p_abs < c->base_offset) break;
        size_t off = abs_pos - p_abs;
        if (off > MAX_OFFSET) break;
        const uint8_t *m = data + (p_abs - c->base_offset);
        if (pos + best_len >= len) break;
        if (!(best_len >= 3 && m[best_len] != src[best_len])) {
            if (safe4) {
                uint32_t m4;
                memcpy(&m4, m, 4);
                if (m4 == src4) {
                    unsigned maxl = len - pos < MAX_MATCH ? (unsigned)(len - pos) : MAX_MATCH;
                    unsigned l = match_len(m, src, maxl);
                    if (l > best_len) {
                        best_len = l;
                        best_off = (unsigned)off;
                        if (l >= c->nice_len || l == MAX_MATCH) break;
                    }
                } else if (best_len < 3 && (m4 & 0xFFFFFF) == src3) {
                    best_len = 3;
                    best_off = (unsigned)off;
                }
            } else if (load24(m) == src3) {
                unsigned maxl = len - pos < MAX_MATCH ? (unsigned)(len - pos) : MAX_MATCH;
                unsigned l = match_len(m, src, maxl);
                if (l > best_len) {
                    best_len = l;
                    best_off = (unsigned)off;
                    if (l >= c->nice_len || l == MAX_MATCH) break;
                }
            }
        }
        uint16_t d = c->prev_tab[p_abs & (WINDOW_SIZE - 1)];
        if (d == 0) break;
        cur -= d;
        depth++;
    }
    *offset_ret = best_off;
    return best_len;
}

/* matchfinder.rs:1059-1106 */
static void hc_skip_positions(orc_compressor *c, const uint8_t *data, size_t len, size_t pos,
                              size_t count)
{
    for (size_t i = 0; i < count; i++)
        if (pos + i + 3 <= len) hc_insert(c, data, pos + i);
}

/* single-probe table, level 1; matchfinder.rs:1139-1208 */
static unsigned ht_find_match(orc_compressor *c, const uint8_t *data, size_t len, size_t pos,
                              unsigned *offset_ret)
{
    *offset_ret = 0;
    if (pos + 3 > len) return 0;
    const uint8_t *src = data + pos;
    uint32_t src3 = load24(src);
    uint32_t h = hash3(src3);
    size_t abs_pos = c->base_offset + pos;
    int32_t cur = c->hash_tab[h];
    c->hash_tab[h] = (int32_t)abs_pos;
    if (cur == -1 || (size_t)cur < c->base_offset) return 0;
    size_t off = abs_pos - (size_t)cur;
    if (off > MAX_OFFSET) return 0;
    const uint8_t *m = data + ((size_t)cur - c->base_offset);
    if (load24(m) != src3) return 0;
    unsigned maxl = len - pos < MAX_MATCH ? (unsigned)(len - pos) : MAX_MATCH;
    *offset_ret = (unsigned)off;
    return match_len(m, src, maxl);
}

/* test hook: huffman_comp.rs:125-155 on a caller's frequencies (lens and codewords hold num_syms entries) */
void orc_make_huffman_code(unsigned num_syms, unsigned max_len, const uint32_t *freqs, uint8_t *lens,
                           uint32_t *codewords)
{
    make_huffman_code(num_syms, max_len, freqs, lens, codewords);
}

void orc_hc_find_match_kat(const uint8_t *data, size_t len, size_t pos, int max_depth,
                           int nice_len, int *out_len, int *out_offset)
{
    orc_compressor *c = orc_compressor_new(6);
    c->max_depth = (unsigned)max_depth;
    c->nice_len = (unsigned)nice_len;
    hc_skip_positions(c, data, len, 0, pos);
    unsigned off;
    *out_len = (int)hc_find_match(c, data, len, pos, &off);
    *out_offset = (int)off;
    orc_compressor_free(c);
}

/* ---- binary-tree match finder (src/compress/matchfinder.rs:1308-1776) ---- */

enum { VIS_BEST, VIS_ALL, VIS_NONE };
typedef struct {
    int kind;
    unsigned best_len, best_off; /* VIS_BEST: matchfinder.rs:1239-1267; VIS_ALL starts at 3 (:1274-1281) */
    uint16_t (*matches)[2];
    unsigned nmatches;
} bt_visitor;

static inline void vis_hash3(bt_visitor *v, unsigned len, unsigned off)
{
    if (v->kind == VIS_BEST) { /* matchfinder.rs:1255-1258 */
        v->best_len = len;
        v->best_off = off;
    } else if (v->kind == VIS_ALL) { /* :1285-1287 */
        v->matches[v->nmatches][0] = (uint16_t)len;
        v->matches[v->nmatches][1] = (uint16_t)off;
        v->nmatches++;
    }
}

static inline void vis_match(bt_visitor *v, unsigned len, unsigned off)
{
    if (v->kind == VIS_NONE) return;
    if (len > v->best_len) { /* :1261-1266, :1290-1295 */
        v->best_len = len;
        v->best_off = off;
        if (v->kind == VIS_ALL) {
            v->matches[v->nmatches][0] = (uint16_t)len;
            v->matches[v->nmatches][1] = (uint16_t)off;
            v->nmatches++;
        }
    }
}

static void bt_reset(orc_compressor *c) /* matchfinder.rs:1327-1331 */
{
    memset(c->bt_hash3, 0xFF, BT_HASH_SIZE * sizeof(*c->bt_hash3));
    memset(c->bt_hash4, 0xFF, BT_HASH_SIZE * sizeof(int32_t));
    c->base_offset = 0;
}

/* matchfinder.rs:1344-1463 */
static void bt_advance_one_byte(orc_compressor *c, const uint8_t *data, size_t data_len, size_t pos,
                                unsigned max_len, unsigned nice_len, unsigned max_depth,
                                bt_visitor *v)
{
    if (pos + 4 > data_len) return;
    const uint8_t *src = data + pos;
    uint32_t val = src[0] | (src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
    uint32_t h3 = ((val & 0xFFFFFFu) * 0x1E35A7BDu) >> 16;
    uint32_t h4 = (val * 0x1E35A7BDu) >> 16;
    size_t abs_pos = c->base_offset + pos;

    int32_t cur_node_3 = c->bt_hash3[h3][0];
    c->bt_hash3[h3][0] = (int32_t)abs_pos;
    int32_t cur_node_3_2 = c->bt_hash3[h3][1];
    c->bt_hash3[h3][1] = cur_node_3;

    int32_t cutoff = (int32_t)abs_pos - (int32_t)WINDOW_SIZE;

    if (cur_node_3 != -1 && cur_node_3 > cutoff && (size_t)cur_node_3 >= c->base_offset) {
        const uint8_t *m = data + ((size_t)cur_node_3 - c->base_offset);
        if (m[0] == src[0] && m[1] == src[1] && m[2] == src[2]) {
            vis_hash3(v, 3, (unsigned)(abs_pos - (size_t)cur_node_3));
        } else if (cur_node_3_2 != -1 && cur_node_3_2 > cutoff &&
                   (size_t)cur_node_3_2 >= c->base_offset) {
            const uint8_t *m2 = data + ((size_t)cur_node_3_2 - c->base_offset);
            if (m2[0] == src[0] && m2[1] == src[1] && m2[2] == src[2])
                vis_hash3(v, 3, (unsigned)(abs_pos - (size_t)cur_node_3_2));
        }
    }

    int32_t cur_node = c->bt_hash4[h4];
    c->bt_hash4[h4] = (int32_t)abs_pos;
    size_t child_idx = abs_pos & (WINDOW_SIZE - 1);
    if (cur_node == -1 || cur_node <= cutoff || (size_t)cur_node < c->base_offset) {
        c->bt_child[child_idx][0] = -1;
        c->bt_child[child_idx][1] = -1;
        return;
    }
    unsigned depth_remaining = max_depth;
    size_t pending_lt_node = child_idx, pending_gt_node = child_idx;
    int pending_lt_child = 0, pending_gt_child = 1;
    unsigned max_len_clamped = max_len < data_len - pos ? max_len : (unsigned)(data_len - pos);
    for (;;) {
        size_t p_abs = (size_t)cur_node;
        size_t p_child_idx = p_abs & (WINDOW_SIZE - 1);
        const uint8_t *m = data + (p_abs - c->base_offset);
        unsigned len = match_len(m, src, max_len_clamped);
        vis_match(v, len, (unsigned)(abs_pos - p_abs));
        if (len >= nice_len || len == max_len_clamped) {
            int32_t c0 = c->bt_child[p_child_idx][0], c1 = c->bt_child[p_child_idx][1];
            c->bt_child[pending_lt_node][pending_lt_child] = c0;
            c->bt_child[pending_gt_node][pending_gt_child] = c1;
            return;
        }
        if (len < max_len_clamped && m[len] < src[len]) {
            c->bt_child[pending_lt_node][pending_lt_child] = cur_node;
            pending_lt_node = p_child_idx;
            pending_lt_child = 1;
            cur_node = c->bt_child[p_child_idx][1];
        } else {
            c->bt_child[pending_gt_node][pending_gt_child] = cur_node;
            pending_gt_node = p_child_idx;
            pending_gt_child = 0;
            cur_node = c->bt_child[p_child_idx][0];
        }
        if (cur_node == -1 || cur_node <= cutoff || (size_t)cur_node < c->base_offset ||
            --depth_remaining == 0) {
            c->bt_child[pending_lt_node][pending_lt_child] = -1;
            c->bt_child[pending_gt_node][pending_gt_child] = -1;
            return;
        }
    }
}

static unsigned bt_find_match(orc_compressor *c, const uint8_t *data, size_t data_len, size_t pos,
                              unsigned *off) /* matchfinder.rs:1465-1531 */
{
    bt_visitor v = {VIS_BEST, 0, 0, NULL, 0};
    bt_advance_one_byte(c, data, data_len, pos, MAX_MATCH, c->nice_len, c->max_depth, &v);
    *off = v.best_off;
    return v.best_len;
}

static void bt_skip_positions(orc_compressor *c, const uint8_t *data, size_t data_len, size_t pos,
                              size_t count) /* matchfinder.rs:1606-1701 */
{
    bt_visitor v = {VIS_NONE, 0, 0, NULL, 0};
    for (size_t k = 0; k < count; k++)
        bt_advance_one_byte(c, data, data_len, pos + k, MAX_MATCH, c->nice_len, c->max_depth, &v);
}

static unsigned bt_find_matches(orc_compressor *c, const uint8_t *data, size_t data_len, size_t pos)
{ /* matchfinder.rs:1703-1775; returns the number of matches left in c->matches */
    bt_visitor v = {VIS_ALL, 3, 0, c->matches, 0};
    bt_advance_one_byte(c, data, data_len, pos, MAX_MATCH, c->nice_len, c->max_depth, &v);
    return v.nmatches;
}

/* ---- parsing ------------------------------------------------------------- */

/* src/compress/mod.rs:1261-1373 */
static size_t decide_greedy_sequences(orc_compressor *c, const uint8_t *in, size_t n,
                                      size_t start, unsigned lazy_depth)
{
    c->nseq = 0;
    memset(&c->stats, 0, sizeof(c->stats));
    memset(c->litlen_freqs, 0, sizeof(c->litlen_freqs));
    memset(c->offset_freqs, 0, sizeof(c->offset_freqs));
    uint32_t litrunlen = 0;
    size_t i = start;
#define LITERAL()                          \
    do {                                   \
        observe_literal(&c->stats, in[i]); \
        c->litlen_freqs[in[i]]++;          \
        litrunlen++;                       \
        i++;                               \
    } while (0)
    while (i < n) {
        if (should_end_block(&c->stats, i - start, n - i)) break;
        unsigned off, len = hc_find_match(c, in, n, i, &off);
        if (len >= 3) {
            unsigned skipped = 0;
            if (lazy_depth >= 1 && i + 1 < n && len < c->nice_len) {
                unsigned off1, len1 = hc_find_match(c, in, n, i + 1, &off1);
                if (len1 > len) {
                    if (lazy_depth >= 2 && i + 2 < n) {
                        unsigned off2, len2 = hc_find_match(c, in, n, i + 2, &off2);
                        if (len2 > len1) {
                            LITERAL();
                            LITERAL();
                            len = len2;
                            off = off2;
                        } else {
                            LITERAL();
                            len = len1;
                            off = off1;
                            skipped = 1;
                        }
                    } else {
                        LITERAL();
                        len = len1;
                        off = off1;
                    }
                } else {
                    skipped = 1;
                }
            }
            unsigned slot = offset_slot(off);
            push_seq(c, litrunlen, len, off, slot);
            observe_match(&c->stats, len, slot);
            c->litlen_freqs[257 + length_slot_tab[len]]++;
            c->offset_freqs[slot]++;
            litrunlen = 0;
            if (len - 1 > skipped)
                hc_skip_positions(c, in, n, i + 1 + skipped, len - 1 - skipped);
            i += len;
        } else {
            LITERAL();
        }
    }
#undef LITERAL
    push_seq(c, litrunlen, 0, 0, 0);
    c->litlen_freqs[256]++;
    return i - start;
}

/* ---- block output -------------------------------------------------------- */

static const uint8_t precode_perm[19] = {16, 17, 18, 0, 8,  7, 9,  6, 10, 5,
                                         11, 4,  12, 3, 13, 2, 14, 1, 15};

/* src/compress/mod.rs:1775-1883 */
static void write_dynamic_header(const orc_compressor *c, bitw *w)
{
    unsigned nl = NUM_LITLEN, no = NUM_OFFSET;
    while (nl > 257 && c->litlen_lens[nl - 1] == 0) nl--;
    while (no > 1 && c->offset_lens[no - 1] == 0) no--;
    bw_put(w, nl - 257, 5);
    bw_put(w, no - 1, 5);
    uint8_t lens[NUM_LITLEN + NUM_OFFSET];
    unsigned total = nl + no;
    memcpy(lens, c->litlen_lens, nl);
    memcpy(lens + nl, c->offset_lens, no);

    uint32_t pre_freqs[19] = {0};
    uint16_t items[NUM_LITLEN + NUM_OFFSET];
    unsigned nitems = 0;
    for (unsigned i = 0; i < total;) {
        unsigned len = lens[i], run = 1;
        while (i + run < total && lens[i + run] == len) run++;
        unsigned total_run = run;
        if (len == 0) {
            while (run >= 11) {
                unsigned k = run < 138 ? run : 138;
                items[nitems++] = (uint16_t)((18 << 8) | (k - 11));
                pre_freqs[18]++;
                run -= k;
            }
            if (run >= 3) {
                unsigned k = run < 10 ? run : 10;
                items[nitems++] = (uint16_t)((17 << 8) | (k - 3));
                pre_freqs[17]++;
                run -= k;
            }
        } else if (run >= 4) {
            items[nitems++] = (uint16_t)(len << 8);
            pre_freqs[len]++;
            run--;
            while (run >= 3) {
                unsigned k = run < 6 ? run : 6;
                items[nitems++] = (uint16_t)((16 << 8) | (k - 3));
                pre_freqs[16]++;
                run -= k;
            }
        }
        while (run > 0) {
            items[nitems++] = (uint16_t)(len << 8);
            pre_freqs[len]++;
            run--;
        }
        i += total_run;
    }
    uint8_t pre_lens[19] = {0};
    uint32_t pre_codes[19] = {0};
    make_huffman_code(19, MAX_PRE_CODEWORD_LEN, pre_freqs, pre_lens, pre_codes);
    unsigned np = 19;
    while (np > 4 && pre_lens[precode_perm[np - 1]] == 0) np--;
    bw_put(w, np - 4, 4);
    for (unsigned j = 0; j < np; j++) bw_put(w, pre_lens[precode_perm[j]], 3);
    for (unsigned j = 0; j < nitems; j++) {
        unsigned sym = items[j] >> 8, extra = items[j] & 0xFF;
        bw_put(w, pre_codes[sym], pre_lens[sym]);
        if (sym == 16) bw_put(w, extra, 2);
        else if (sym == 17) bw_put(w, extra, 3);
        else if (sym == 18) bw_put(w, extra, 7);
    }
}

/* src/compress/mod.rs:1952-2186: literals, then length symbol + extra bits,
 * then offset symbol + extra bits, per sequence */
static void write_sequences(const orc_compressor *c, bitw *w, const uint8_t *in, size_t start)
{
    size_t pos = start;
    for (size_t s = 0; s < c->nseq; s++) {
        const sequence *q = &c->seqs[s];
        for (uint32_t k = 0; k < q->litrunlen; k++) {
            unsigned lit = in[pos++];
            bw_put(w, c->litlen_codes[lit], c->litlen_lens[lit]);
        }
        unsigned len = q->length & 0x1FF;
        if (len == 0) continue;
        unsigned lslot = length_slot_tab[len], oslot = q->length >> 9;
        bw_put(w, c->litlen_codes[257 + lslot] | ((len - length_base[lslot]) << c->litlen_lens[257 + lslot]),
               c->litlen_lens[257 + lslot] + length_extra[lslot]);
        bw_put(w, c->offset_codes[oslot] | ((q->offset - offset_base[oslot]) << c->offset_lens[oslot]),
               c->offset_lens[oslot] + offset_extra[oslot]);
        pos += len;
    }
}

/* static code lengths, src/compress/mod.rs:161-234 */
static void load_static_codes(orc_compressor *c)
{
    unsigned i = 0;
    for (; i < 144; i++) c->litlen_lens[i] = 8;
    for (; i < 256; i++) c->litlen_lens[i] = 9;
    for (; i < 280; i++) c->litlen_lens[i] = 7;
    for (; i < 288; i++) c->litlen_lens[i] = 8;
    for (i = 0; i < 32; i++) c->offset_lens[i] = 5;
    /* canonical codewords from the lengths (mod.rs:131-151) */
    unsigned counts[16] = {0}, next[16] = {0};
    for (i = 0; i < 288; i++) counts[c->litlen_lens[i]]++;
    unsigned code = 0;
    for (unsigned l = 1; l <= 9; l++) {
        code = (code + counts[l - 1]) << 1;
        next[l] = code;
    }
    next[0] = 0;
    for (i = 0; i < 288; i++)
        c->litlen_codes[i] = reverse_codeword(next[c->litlen_lens[i]]++, c->litlen_lens[i]);
    for (i = 0; i < 32; i++) c->offset_codes[i] = reverse_codeword(i, 5);
}

/* src/compress/mod.rs:1466-1584.  Returns bytes consumed, 0 on overflow. */
static size_t compress_greedy_block(orc_compressor *c, const uint8_t *in, size_t n, size_t start,
                                    bitw *w, unsigned lazy_depth, int final_block)
{
    size_t processed;
    if (c->level >= 2) {
        processed = decide_greedy_sequences(c, in, n, start, lazy_depth);
        int is_final = (start + processed >= n) && final_block;
        make_huffman_code(NUM_LITLEN, MAX_LITLEN_CODEWORD_LEN, c->litlen_freqs, c->litlen_lens,
                          c->litlen_codes);
        make_huffman_code(NUM_OFFSET, MAX_OFFSET_CODEWORD_LEN, c->offset_freqs, c->offset_lens,
                          c->offset_codes);
        bw_put(w, is_final, 1);
        bw_put(w, 2, 2);
        write_dynamic_header(c, w);
    } else {
        /* level 1: static Huffman codes, single-probe table (mod.rs:1499-1583) */
        load_static_codes(c);
        c->nseq = 0;
        memset(&c->stats, 0, sizeof(c->stats));
        uint32_t litrunlen = 0;
        size_t i = start;
        int split = n > 65536;
        while (i < n) {
            if (split && should_end_block(&c->stats, i - start, n - i)) break;
            unsigned off, len = ht_find_match(c, in, n, i, &off);
            if (len >= 3) {
                unsigned slot = offset_slot(off);
                if (split) observe_match(&c->stats, len, slot);
                push_seq(c, litrunlen, len, off, slot);
                litrunlen = 0;
                i += len; /* skipped positions are not inserted (matchfinder.rs:1231) */
            } else {
                if (split) observe_literal(&c->stats, in[i]);
                litrunlen++;
                i++;
            }
        }
        push_seq(c, litrunlen, 0, 0, 0);
        processed = i - start;
        int is_final = (start + processed >= n) && final_block;
        bw_put(w, is_final, 1);
        bw_put(w, 1, 2);
    }
    write_sequences(c, w, in, start);
    bw_put(w, c->litlen_codes[256], c->litlen_lens[256]);
    if (w->overflow) return 0;
    /* an empty input still emits a block; report progress through a flag */
    return processed;
}

/* src/compress/mod.rs:1586-1773.  Returns bytes consumed, 0 on overflow. */
static size_t compress_near_optimal_block(orc_compressor *c, const uint8_t *in, size_t n,
                                          size_t start, bitw *w, int final_block)
{
    /* pass 1: greedy parse with the binary tree, for the block end and first statistics */
    memset(&c->stats, 0, sizeof(c->stats));
    memset(c->litlen_freqs, 0, sizeof(c->litlen_freqs));
    memset(c->offset_freqs, 0, sizeof(c->offset_freqs));
    size_t in_idx = start;
    while (in_idx < n) {
        if (should_end_block(&c->stats, in_idx - start, n - in_idx)) break;
        unsigned off, len = bt_find_match(c, in, n, in_idx, &off);
        if (len >= 3) {
            unsigned slot = offset_slot(off);
            observe_match(&c->stats, len, slot);
            c->litlen_freqs[257 + length_slot_tab[len]]++;
            c->offset_freqs[slot]++;
            bt_skip_positions(c, in, n, in_idx + 1, len - 1);
            in_idx += len;
        } else {
            observe_literal(&c->stats, in[in_idx]);
            c->litlen_freqs[in[in_idx]]++;
            in_idx++;
        }
    }
    const size_t processed = in_idx - start;
    const uint8_t *block = in + start;
    const int is_final = (start + processed >= n) && final_block;

    c->nseq = 0;
    c->litlen_freqs[256]++;
    make_huffman_code(NUM_LITLEN, MAX_LITLEN_CODEWORD_LEN, c->litlen_freqs, c->litlen_lens,
                      c->litlen_codes);
    make_huffman_code(NUM_OFFSET, MAX_OFFSET_CODEWORD_LEN, c->offset_freqs, c->offset_lens,
                      c->offset_codes);
    /* update_costs, mod.rs:2209-2224: unused symbols cost 0 bits, as in the reference */
    for (unsigned len = 3; len <= MAX_MATCH; len++) {
        unsigned ls = length_slot_tab[len];
        c->length_costs[len] = c->litlen_lens[257 + ls] + length_extra[ls];
    }
    for (unsigned slot = 0; slot < 30; slot++)
        c->offset_slot_costs[slot] = c->offset_lens[slot] + offset_extra[slot];

    for (size_t i = 0; i <= processed; i++) c->dp_costs[i] = 0x3FFFFFFFu;
    c->dp_costs[0] = 0;

    /* pass 2: forward relaxation over the block slice with a fresh match finder */
    bt_reset(c);
    size_t pos = 0;
    while (pos < processed) {
        uint32_t cur = c->dp_costs[pos];
        if (cur >= 0x3FFFFFFFu) {
            pos++;
            continue;
        }
        uint32_t lit = c->litlen_lens[block[pos]];
        if (cur + lit < c->dp_costs[pos + 1]) {
            c->dp_costs[pos + 1] = cur + lit;
            c->dp_path[pos + 1] = 1;
        }
        unsigned nm = bt_find_matches(c, block, processed, pos);
        unsigned best_len = 0;
        for (unsigned k = 0; k < nm; k++) {
            unsigned len = c->matches[k][0], off = c->matches[k][1];
            if (len > best_len) best_len = len;
            unsigned real_off = off; /* 1..32768 fits the reference's u16 */
            uint32_t cost = c->length_costs[len] + c->offset_slot_costs[offset_slot(real_off)];
            if (cur + cost < c->dp_costs[pos + len]) {
                c->dp_costs[pos + len] = cur + cost;
                c->dp_path[pos + len] = len | ((uint32_t)real_off << 16);
            }
        }
        if (best_len >= c->nice_len) {
            bt_skip_positions(c, block, processed, pos + 1, best_len - 1);
            pos += best_len;
        } else {
            pos++;
        }
    }

    /* backtrack, then rebuild statistics and sequences in forward order (mod.rs:1719-1751) */
    memset(c->litlen_freqs, 0, sizeof(c->litlen_freqs));
    memset(c->offset_freqs, 0, sizeof(c->offset_freqs));
    c->litlen_freqs[256] = 1;
    size_t nnodes = 0;
    for (size_t p = processed; p > 0;) {
        uint32_t packed = c->dp_path[p];
        /* node list reuses dp_costs from the top: every cost below p is dead once visited */
        c->dp_costs[processed - nnodes] = packed;
        nnodes++;
        p -= packed & 0xFFFF;
    }
    uint32_t litrunlen = 0;
    size_t cur_pos = 0;
    for (size_t k = nnodes; k-- > 0;) {
        uint32_t packed = c->dp_costs[processed - k];
        unsigned length = packed & 0xFFFF, offset = packed >> 16;
        if (length == 1) {
            c->litlen_freqs[block[cur_pos]]++;
            litrunlen++;
            cur_pos++;
        } else {
            unsigned slot = offset_slot(offset);
            push_seq(c, litrunlen, length, offset, slot);
            c->litlen_freqs[257 + length_slot_tab[length]]++;
            c->offset_freqs[slot]++;
            litrunlen = 0;
            cur_pos += length;
        }
    }
    push_seq(c, litrunlen, 0, 0, 0);
    make_huffman_code(NUM_LITLEN, MAX_LITLEN_CODEWORD_LEN, c->litlen_freqs, c->litlen_lens,
                      c->litlen_codes);
    make_huffman_code(NUM_OFFSET, MAX_OFFSET_CODEWORD_LEN, c->offset_freqs, c->offset_lens,
                      c->offset_codes);
    /* write_dynamic_block_with_sequences, mod.rs:1375-1398 */
    bw_put(w, is_final, 1);
    bw_put(w, 2, 2);
    write_dynamic_header(c, w);
    write_sequences(c, w, in, start);
    bw_put(w, c->litlen_codes[256], c->litlen_lens[256]);
    if (w->overflow) return 0;
    return processed;
}

/* src/compress/mod.rs:1400-1464 */
static int compress_uncompressed(const uint8_t *in, size_t n, uint8_t *out, size_t cap, int flush,
                                 size_t *out_len)
{
    size_t o = 0, i = 0;
    while (i < n) {
        size_t blen = n - i < 65535 ? n - i : 65535;
        int bfinal = (i + 65535 >= n) && flush == ORC_FLUSH_FINISH;
        if (o + 5 + blen > cap) return 1;
        out[o++] = (uint8_t)bfinal;
        out[o++] = (uint8_t)blen;
        out[o++] = (uint8_t)(blen >> 8);
        out[o++] = (uint8_t)~blen;
        out[o++] = (uint8_t)(~blen >> 8);
        memcpy(out + o, in + i, blen);
        o += blen;
        i += blen;
    }
    if (flush == ORC_FLUSH_SYNC) {
        if (o + 5 > cap) return 1;
        out[o++] = 0;
        out[o++] = 0;
        out[o++] = 0;
        out[o++] = 0xFF;
        out[o++] = 0xFF;
    }
    *out_len = o;
    return 0;
}

/* src/compress/mod.rs:604-691 for one chunk (<= 256 KiB) */
static int compress_chunk(orc_compressor *c, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                          int flush, size_t *out_len)
{
    if (c->level == 0) return compress_uncompressed(in, n, out, cap, flush, out_len);
    bitw w = {out, cap, 0, 0, 0, 0};
    /* matchfinder.rs:743-747, :1333-1337 */
    if (c->base_offset + n > 0x7FFFFFFFu) {
        if (c->level >= 10) bt_reset(c);
        else memset(c->hash_tab, 0xFF, HASH_SIZE * sizeof(int32_t));
        c->base_offset = 0;
    }
    unsigned lazy = c->level >= 8 ? 2 : c->level >= 5 ? 1 : 0;
    size_t i = 0;
    int fail = 0;
    while (i < n) {
        size_t p = c->level >= 10
                       ? compress_near_optimal_block(c, in, n, i, &w, flush == ORC_FLUSH_FINISH)
                       : compress_greedy_block(c, in, n, i, &w, lazy, flush == ORC_FLUSH_FINISH);
        if (p == 0) {
            fail = 1;
            break;
        }
        i += p;
    }
    if (!fail && n == 0 && flush == ORC_FLUSH_FINISH) {
        if (c->level >= 10) compress_near_optimal_block(c, in, 0, 0, &w, 1);
        else compress_greedy_block(c, in, 0, 0, &w, 0, 1);
        if (w.overflow) fail = 1;
    }
    if (!fail && flush == ORC_FLUSH_SYNC) {
        /* mod.rs:662-681: empty stored block = 3 header bits, align, 00 00 FF FF */
        bw_put(&w, 0, 3);
        bw_flush(&w);
        if (w.idx + 4 > cap) {
            fail = 1;
        } else {
            out[w.idx++] = 0;
            out[w.idx++] = 0;
            out[w.idx++] = 0xFF;
            out[w.idx++] = 0xFF;
        }
    }
    bw_flush(&w);
    c->base_offset += n; /* matchfinder.rs:749-751 */
    if (fail || w.overflow) return 1;
    *out_len = w.idx;
    return 0;
}

size_t orc_compress_bound(int format, size_t n)
{
    size_t b = n + (n / 65535 + 1) * 5 + 10;
    if (format == ORC_FMT_ZLIB) b += 6;
    if (format == ORC_FMT_GZIP) b += 18;
    return b;
}

/* src/compress/mod.rs:693-790 */
int orc_compress_raw(orc_compressor *c, const uint8_t *in, size_t n, uint8_t *out, size_t cap,
                     int flush, size_t *out_len)
{
    if (n <= CHUNK_SIZE) return compress_chunk(c, in, n, out, cap, flush, out_len);
    /* each chunk is compressed on its own into a bound-sized scratch buffer and
     * the results are concatenated (mod.rs:699-772) */
    size_t o = 0;
    size_t sbound = orc_compress_bound(ORC_FMT_DEFLATE, CHUNK_SIZE);
    uint8_t *scratch = malloc(sbound);
    if (!scratch) return 1;
    for (size_t i = 0; i < n; i += CHUNK_SIZE) {
        size_t clen = n - i < CHUNK_SIZE ? n - i : CHUNK_SIZE;
        int last = i + clen >= n;
        size_t got = 0;
        if (compress_chunk(c, in + i, clen, scratch, orc_compress_bound(ORC_FMT_DEFLATE, clen),
                           last ? flush : ORC_FLUSH_SYNC, &got) ||
            o + got > cap) {
            free(scratch);
            return 1;
        }
        memcpy(out + o, scratch, got);
        o += got;
    }
    free(scratch);
    *out_len = o;
    return 0;
}

int orc_compress(orc_compressor *c, int format, const uint8_t *in, size_t n, uint8_t *out,
                 size_t cap, size_t *out_len)
{
    size_t body = 0;
    *out_len = 0;
    if (format == ORC_FMT_DEFLATE) {
        if (orc_compress_raw(c, in, n, out, cap, ORC_FLUSH_FINISH, &body)) return 1;
        *out_len = body;
        return 0;
    }
    if (format == ORC_FMT_ZLIB) {
        /* src/compress/mod.rs:2248-2298 */
        if (cap < 6) return 1;
        unsigned hdr = (8u << 8) | (7u << 12);
        unsigned hint = c->level < 2 ? 0 : c->level < 6 ? 1 : c->level < 8 ? 2 : 3;
        hdr |= hint << 6;
        hdr |= 31 - (hdr % 31);
        out[0] = (uint8_t)(hdr >> 8);
        out[1] = (uint8_t)hdr;
        if (orc_compress_raw(c, in, n, out + 2, cap - 6, ORC_FLUSH_FINISH, &body)) return 1;
        uint32_t a = orc_adler32(1, in, n);
        uint8_t *f = out + 2 + body;
        f[0] = (uint8_t)(a >> 24);
        f[1] = (uint8_t)(a >> 16);
        f[2] = (uint8_t)(a >> 8);
        f[3] = (uint8_t)a;
        *out_len = body + 6;
        return 0;
    }
    /* src/compress/mod.rs:2300-2357 */
    if (cap < 18) return 1;
    static const uint8_t h[4] = {0x1F, 0x8B, 8, 0};
    memcpy(out, h, 4);
    memset(out + 4, 0, 4);
    out[8] = c->level < 2 ? 4 : c->level >= 8 ? 2 : 0;
    out[9] = 255;
    if (orc_compress_raw(c, in, n, out + 10, cap - 18, ORC_FLUSH_FINISH, &body)) return 1;
    uint32_t crc = orc_crc32(0, in, n);
    uint32_t isz = (uint32_t)n;
    uint8_t *f = out + 10 + body;
    for (int k = 0; k < 4; k++) {
        f[k] = (uint8_t)(crc >> (8 * k));
        f[4 + k] = (uint8_t)(isz >> (8 * k));
    }
    *out_len = body + 18;
    return 0;
}
