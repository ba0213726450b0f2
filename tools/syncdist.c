/* Development tool: measures how many bits a DEFLATE decoder started at an
 * arbitrary bit position needs before its token boundaries coincide with the
 * true ones (self-synchronisation distance).  Includes the oracle's inflater
 * source to reuse its tables.  Not part of the product or the tests. */
#include "../oracle/inflate.c"
#include <stdio.h>
#include <stdlib.h>

static uint64_t peek(const uint8_t *in, size_t n, uint64_t bit)
{
    uint64_t v = 0;
    size_t byte = bit >> 3;
    for (int k = 0; k < 8; k++)
        if (byte + k < n) v |= (uint64_t)in[byte + k] << (8 * k);
    return v >> (bit & 7);
}

/* decode one token at bit position; returns bits consumed, *kind 0 lit 1 match 2 eob */
static unsigned token(inflater *d, const uint8_t *in, size_t n, uint64_t bit, int *kind)
{
    uint64_t bits = peek(in, n, bit);
    uint32_t e = d->litlen_table[bits & ((1u << d->litlen_tablebits) - 1)];
    unsigned used = 0;
    if ((e & HUFFDEC_EXCEPTIONAL) && (e & HUFFDEC_SUBTABLE_POINTER)) {
        used = e & 0xFF;
        bits >>= used;
        e = d->litlen_table[(e >> 16) + (bits & ((1u << ((e >> 8) & 0x3F)) - 1))];
    }
    used += e & 0xFF;
    if (e & HUFFDEC_LITERAL) { *kind = 0; return used; }
    if (e & HUFFDEC_END_OF_BLOCK) { *kind = 2; return used; }
    *kind = 1;
    bits = peek(in, n, bit + used);
    e = d->offset_table[bits & 0xFF];
    if (e & HUFFDEC_SUBTABLE_POINTER) {
        unsigned m = e & 0xFF;
        used += m;
        bits >>= m;
        e = d->offset_table[(e >> 16) + (bits & ((1u << ((e >> 8) & 0x3F)) - 1))];
    }
    return used + (e & 0xFF);
}

int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END);
    size_t n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *in = malloc(n + 16);
    if (fread(in, 1, n, f) != n) return 1;
    unsigned S = argc > 2 ? atoi(argv[2]) : 128;
    inflater d;
    bitreader b = {in, n, 0, 0, 0};
    uint8_t *mark = calloc(n * 8 + 64, 1);
    unsigned long hist[64] = {0}, total = 0;
    unsigned long long sumd = 0;
    for (;;) {
        refill(&b);
        int fin = b.bitbuf & 1;
        unsigned type = (b.bitbuf >> 1) & 3;
        TAKE(&b, 3);
        if (type != 2) { fprintf(stderr, "non-dynamic block, stop\n"); break; }
        if (read_dynamic_header(&d, &b) != 0) return 2;
        uint64_t start = b.in_idx * 8 - b.bitsleft, pos = start;
        /* true chain */
        for (;;) {
            int kind;
            mark[pos] = 1;
            pos += token(&d, in, n, pos, &kind);
            if (kind == 2) break;
        }
        uint64_t end = pos;
        /* wrong starts at every S-bit boundary inside the block */
        for (uint64_t s = start + S; s + 64 < end; s += S) {
            uint64_t p = s;
            while (p < end && !mark[p]) {
                int kind;
                p += token(&d, in, n, p, &kind);
                if (kind == 2) break;
            }
            unsigned long dist = (unsigned long)((p - s) / S);
            hist[dist > 63 ? 63 : dist]++;
            sumd += p - s;
            total++;
        }
        /* advance the real reader to the end of block */
        b.in_idx = end >> 3; b.bitbuf = 0; b.bitsleft = 0;
        refill(&b); TAKE(&b, end & 7);
        if (fin) break;
    }
    printf("S=%u samples=%lu mean sync distance %.1f bits\n", S, total, (double)sumd / total);
    unsigned long cum = 0;
    for (int i = 0; i < 64; i++) {
        cum += hist[i];
        if (hist[i]) printf("  d=%2d sub-sequences: %6lu  (cum %.4f)\n", i, hist[i], (double)cum / total);
    }
    return 0;
}
