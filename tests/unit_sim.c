/* How deep are the dependencies between the matches of a DEFLATE stream when its output is produced in groups
 * and every match is cut into pieces of at most 16 bytes that are copied as soon as their source bytes are
 * final?  tests/unit_sim <raw deflate file> <output size> [group bytes] [unit cap]: rounds per group, pieces
 * visited per round.  (CPU; a model of a copy phase that was considered for the expand kernel.) */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { uint32_t o, len, dist; } match_t;
static match_t *g_m; static size_t g_nm, g_cap;
static void on_match(size_t o, size_t len, size_t dist)
{
    if (g_nm == g_cap) { g_cap = g_cap ? 2 * g_cap : 1 << 20; g_m = realloc(g_m, g_cap * sizeof *g_m); }
    g_m[g_nm++] = (match_t){(uint32_t)o, (uint32_t)len, (uint32_t)dist};
}
#define ORC_ON_MATCH(o, length, offset) on_match(o, length, offset)
#include "../oracle/inflate.c"

typedef struct { uint32_t dst, len, sd; } unit_t;
int main(int argc, char **argv)
{
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    uint8_t *in = malloc(n); if (fread(in, 1, n, f) != (size_t)n) return 1;
    size_t out_len = strtoul(argv[2], 0, 0), ic, op;
    uint32_t G = argc > 3 ? atoi(argv[3]) : 8176, Q = argc > 4 ? atoi(argv[4]) : 1536;
    uint8_t *out = malloc(out_len);
    int r = inflate_raw(in, n, out, out_len, &ic, &op);
    printf("inflate %d, %zu bytes, %zu matches\n", r, op, g_nm);
    uint8_t *pend = calloc(op + 64, 1);
    unit_t *u = malloc(sizeof *u * (Q + 64)), *v = malloc(sizeof *u * (Q + 64));
    size_t mi = 0; uint64_t groups = 0, rounds = 0, visits = 0, units = 0, passes1 = 0, passes4 = 0, maxr = 0;
    uint64_t hist[64] = {0};
    uint32_t gstart = 0;
    while (gstart < op) {
        /* a group: matches that start in [gstart, gstart + G), at most Q pieces */
        uint32_t nu = 0, gend = gstart + G < op ? gstart + G : (uint32_t)op;
        while (mi < g_nm && g_m[mi].o < gend) {
            uint32_t k = (g_m[mi].len + 15) / 16;
            if (nu + k > Q || g_m[mi].o + g_m[mi].len > gstart + G) { gend = g_m[mi].o; break; }
            for (uint32_t j = 0; j < k; j++) {
                uint32_t l = g_m[mi].len - 16 * j; if (l > 16) l = 16;
                u[nu++] = (unit_t){g_m[mi].o + 16 * j, l, g_m[mi].dist};
                memset(pend + g_m[mi].o + 16 * j, 1, l);
            }
            mi++;
        }
        if (gend == gstart) { fprintf(stderr, "stuck\n"); return 1; }
        units += nu; groups++;
        uint32_t rr = 0, left = nu;
        while (left) {
            uint32_t keep = 0;
            visits += left;
            /* four waves while many are left, one wave's passes once few are */
            passes4 += left > 128 ? 4 * ((left / 4 + 63) / 64) : (left + 63) / 64;
            passes1 += (left + 63) / 64;
            uint8_t done[Q + 64];
            for (uint32_t i = 0; i < left; i++) {
                uint32_t need = u[i].sd < u[i].len ? u[i].sd : u[i].len, ok = 1;
                for (uint32_t b = 0; b < need; b++) if (pend[u[i].dst - u[i].sd + b]) { ok = 0; break; }
                done[i] = ok;
            }
            for (uint32_t i = 0; i < left; i++) {
                if (done[i]) memset(pend + u[i].dst, 0, u[i].len); else v[keep++] = u[i];
            }
            unit_t *t = u; u = v; v = t;
            left = keep; rr++;
        }
        rounds += rr; if (rr > maxr) maxr = rr; hist[rr < 63 ? rr : 63]++;
        gstart = gend;
    }
    printf("groups %llu, pieces per group %.0f, rounds per group %.2f (max %llu), piece visits per piece %.3f\n",
           (unsigned long long)groups, (double)units / groups, (double)rounds / groups, (unsigned long long)maxr, (double)visits / units);
    printf("wave passes per group: %.1f (4 waves, then 1), %.1f (ideal)  -> per KiB %.2f\n", (double)passes4 / groups, (double)passes1 / groups,
           (double)passes4 / (op / 1024.0));
    for (int i = 0; i < 64; i++) if (hist[i]) printf("  %d rounds: %llu groups\n", i, (unsigned long long)hist[i]);
    return 0;
}
