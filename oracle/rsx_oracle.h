/*
 * rsx_oracle.h -- CPU restatement of the libdeflate-rsx batch DEFLATE path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * engine: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker / the CPU number printed
 * beside the GPU number.  The product path (librsx_deflate.so) never links,
 * loads or calls it.
 *
 * Parity pinning: the reference is a Rust crate and no Rust toolchain exists in
 * the build image, so the reference cannot be executed.  The restatement is
 * pinned by (a) every known-answer vector the reference's own tests hold for
 * this path (tests/unit_tests.rs, tests/adler32_test.rs, tests/batch_test.rs,
 * src/compress/matchfinder.rs unit tests; copied as data into
 * tests/golden/reference_kat.json) and (b) cross-checks against zlib, which
 * implements the same RFC 1950/1951/1952 functions.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * the reference repository root).
 */
#ifndef RSX_ORACLE_H
#define RSX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DecompressResult, src/decompress/mod.rs:77-85 (declaration order = code). */
enum {
    ORC_SUCCESS = 0,
    ORC_BAD_DATA = 1,
    ORC_SHORT_OUTPUT = 2,
    ORC_INSUFFICIENT_SPACE = 3,
    ORC_SHORT_INPUT = 4,
};

/* container formats */
enum { ORC_FMT_DEFLATE = 0, ORC_FMT_ZLIB = 1, ORC_FMT_GZIP = 2 };

/* FlushMode, src/compress/mod.rs:424-429 */
enum { ORC_FLUSH_NONE = 0, ORC_FLUSH_SYNC = 1, ORC_FLUSH_FINISH = 2 };

/* src/crc32/mod.rs:332-365 ; src/adler32/mod.rs:115-152 */
uint32_t orc_crc32(uint32_t crc, const uint8_t *p, size_t n);
uint32_t orc_adler32(uint32_t adler, const uint8_t *p, size_t n);

/* Compressor::{deflate,zlib,gzip}_compress_bound, src/compress/mod.rs:2236-2246 */
size_t orc_compress_bound(int format, size_t n);

/*
 * One-shot decompression (src/decompress/mod.rs:164-202 for raw DEFLATE,
 * :1074-1127 zlib, :1144-1240 gzip).  Returns a DecompressResult code and
 * fills *in_consumed / *out_produced.
 */
int orc_decompress(int format, const uint8_t *in, size_t in_len, uint8_t *out,
                   size_t out_cap, size_t *in_consumed, size_t *out_produced);

/*
 * One-shot compression (src/compress/mod.rs:693-790 raw, :2248-2298 zlib,
 * :2300-2357 gzip).  Returns 0 = Success, 1 = InsufficientSpace
 * (CompressResult, src/compress/mod.rs:238-241).
 */
typedef struct orc_compressor orc_compressor;
orc_compressor *orc_compressor_new(int level);
void orc_compressor_free(orc_compressor *c);
int orc_compress(orc_compressor *c, int format, const uint8_t *in, size_t in_len,
                 uint8_t *out, size_t out_cap, size_t *out_len);
/* raw DEFLATE with an explicit flush mode (used by the chunked path and tests) */
int orc_compress_raw(orc_compressor *c, const uint8_t *in, size_t in_len, uint8_t *out,
                     size_t out_cap, int flush, size_t *out_len);

/* hash-chain match finder probe for the reference's unit KAT
 * (src/compress/matchfinder.rs:1783-1814): insert positions [0,pos) then
 * search at pos. */
/* make_huffman_code (huffman_comp.rs:125-155): length-limited canonical code for `freqs`; codewords bit-reversed */
void orc_make_huffman_code(unsigned num_syms, unsigned max_len, const uint32_t *freqs, uint8_t *lens,
                           uint32_t *codewords);
void orc_hc_find_match_kat(const uint8_t *data, size_t len, size_t pos, int max_depth,
                           int nice_len, int *out_len, int *out_offset);

/*
 * Batch drivers (src/batch.rs:20-58 and :74-101): flat buffers + offsets, one
 * worker thread per `threads`, per-worker reusable state like rayon map_init.
 * in_off has n+1 entries; out slot i is [out_off[i], out_off[i+1]).
 * status[i]: compress 0/1, decompress = DecompressResult.  out_sizes[i] = 0
 * on failure.
 */
void orc_batch_compress(int format, int level, const uint8_t *in, const uint64_t *in_off,
                        size_t n, uint8_t *out, const uint64_t *out_off,
                        uint64_t *out_sizes, int32_t *status, int threads);
void orc_batch_decompress(int format, const uint8_t *in, const uint64_t *in_off, size_t n,
                          uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                          int32_t *status, int threads);

#ifdef __cplusplus
}
#endif
#endif
