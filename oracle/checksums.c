/*
 * checksums.c -- CRC-32 and Adler-32 restated on the CPU (oracle; test
 * infrastructure only, see rsx_oracle.h).
 */
#include "rsx_oracle.h"

#include <string.h>

/* ---- CRC-32 (IEEE 802.3, reflected polynomial 0xEDB88320) -------------- */

static uint32_t crc_tab[8][256];
static int crc_tab_ready;

/* Table 0 is the byte table of src/crc32_tables.rs:1-3; tables 1..7 are the
 * slice-by-8 extensions used by src/crc32/mod.rs:12-322. */
static void crc_init(void)
{
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int k = 0; k < 8; k++)
            c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
        crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++)
        for (int t = 1; t < 8; t++)
            crc_tab[t][i] = (crc_tab[t - 1][i] >> 8) ^ crc_tab[0][crc_tab[t - 1][i] & 0xFF];
    __atomic_store_n(&crc_tab_ready, 1, __ATOMIC_RELEASE);
}

/* src/crc32/mod.rs:332-365: crc32(crc, buf) = !update(!crc, buf); the byte
 * recurrence is src/crc32/mod.rs:4-9, the 8-byte stride :12-322. */
uint32_t orc_crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    if (!__atomic_load_n(&crc_tab_ready, __ATOMIC_ACQUIRE))
        crc_init();
    uint32_t c = ~crc;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = crc_tab[7][lo & 0xFF] ^ crc_tab[6][(lo >> 8) & 0xFF] ^
            crc_tab[5][(lo >> 16) & 0xFF] ^ crc_tab[4][lo >> 24] ^
            crc_tab[3][hi & 0xFF] ^ crc_tab[2][(hi >> 8) & 0xFF] ^
            crc_tab[1][(hi >> 16) & 0xFF] ^ crc_tab[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--)
        c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xFF];
    return ~c;
}

/* ---- Adler-32 ----------------------------------------------------------- */

#define ADLER_DIVISOR 65521u
/* src/adler32/mod.rs:5 reduces every 4096 bytes (5552 is the largest safe
 * chunk, tests/adler32_test.rs:3). */
#define ADLER_CHUNK 4096u

/* src/adler32/mod.rs:8-104 (generic) behind the dispatcher at :115-152. */
uint32_t orc_adler32(uint32_t adler, const uint8_t *p, size_t n)
{
    uint32_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    while (n) {
        size_t k = n < ADLER_CHUNK ? n : ADLER_CHUNK;
        n -= k;
        while (k >= 4) {
            s1 += p[0]; s2 += s1;
            s1 += p[1]; s2 += s1;
            s1 += p[2]; s2 += s1;
            s1 += p[3]; s2 += s1;
            p += 4;
            k -= 4;
        }
        while (k--) {
            s1 += *p++;
            s2 += s1;
        }
        s1 %= ADLER_DIVISOR;
        s2 %= ADLER_DIVISOR;
    }
    return (s2 << 16) | s1;
}
