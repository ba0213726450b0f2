/*
 * rsx_deflate.h -- C ABI of librsx_deflate.so, the MI355X (gfx950) batch
 * DEFLATE engine that sits behind libdeflate-rsx's batch API.
 *
 * The reference crate exports no C symbols; its only accelerator hook is
 * `CudaBatchCompressor` (src/batch_cuda.rs:20, :43-46), called from
 * `BatchCompressor::compress_batch` (src/batch.rs:21-32).  These entry points
 * are what a `HipBatchCompressor` / `HipBatchDecompressor` in that position
 * binds: the same flat-buffer + u64-offset layout that src/batch_cuda.rs:57-87
 * builds for its kernel (src/cuda/compress.cu:1-8), plain pointers and sizes.
 *
 * Conventions
 *   - Every function returns RSX_OK (0) or a negative RSX_E_* engine error.  An
 *     engine error means "nothing usable was produced": the Rust shim returns
 *     Err and the caller falls back to the CPU path (src/batch.rs:28-30).
 *   - Per-item results are reported in status[i] with the reference's own
 *     codes and out_sizes[i] (0 on failure); one bad item never fails the
 *     batch (src/batch.rs:52-54, :95-97).
 *   - Item i reads  in[in_off[i] .. in_off[i+1])           (n+1 offsets) and
 *     may write     out[out_off[i] .. out_off[i+1])         (n+1 offsets).
 *     Nothing outside slot i is written for item i.
 *   - The library never allocates memory the caller can see.
 *   - An engine is bound to one GPU (rsx_engine_create) or to several
 *     (rsx_engine_create_multi: the host entry points then cut every batch into
 *     one contiguous range of items per GPU).  Calls on one engine serialise on
 *     an internal mutex per GPU; use one engine per thread for concurrency.
 */
#ifndef RSX_DEFLATE_H
#define RSX_DEFLATE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* engine-level return codes */
#define RSX_OK 0
#define RSX_E_INVALID (-1)  /* bad argument (null pointer, level, format) */
#define RSX_E_DEVICE (-2)   /* HIP runtime error; see rsx_engine_last_error */
#define RSX_E_NOMEM (-3)    /* device workspace allocation failed */

/* container formats: raw DEFLATE (batch.rs), zlib, gzip (src/api.rs:22-57) */
#define RSX_FORMAT_DEFLATE 0
#define RSX_FORMAT_ZLIB 1
#define RSX_FORMAT_GZIP 2

/* per-item decompress status = DecompressResult, src/decompress/mod.rs:77-85 */
#define RSX_SUCCESS 0
#define RSX_BAD_DATA 1
#define RSX_SHORT_OUTPUT 2
#define RSX_INSUFFICIENT_SPACE 3
#define RSX_SHORT_INPUT 4
/* not a DecompressResult: the engine stopped making progress on this item (an engine defect, never a
 * property of the data); out_sizes[i] = 0.  A caller treats it like any other failed item. */
#define RSX_INTERNAL_ERROR 100
/* per-item compress status = CompressResult, src/compress/mod.rs:238-241:
 * 0 Success, 1 InsufficientSpace */
#define RSX_COMPRESS_INSUFFICIENT_SPACE 1

/* FlushMode of the raw compressor, src/compress/mod.rs:424-429 (None is not offered: a
 * batch call always ends on a byte boundary) */
#define RSX_FLUSH_SYNC 1   /* end with an empty stored block; the DEFLATE stream stays open */
#define RSX_FLUSH_FINISH 2 /* end with the final block */

/* checksum kinds for rsx_batch_checksum* */
#define RSX_CHECKSUM_CRC32 0   /* src/crc32/mod.rs:332-365 */
#define RSX_CHECKSUM_ADLER32 1 /* src/adler32/mod.rs:115-152 */

typedef struct rsx_engine rsx_engine;

/*
 * Replaces CudaBatchCompressor::new (src/batch_cuda.rs:20-41, which opens
 * device 0): binds an engine to GPU `device`.  `stream` is a hipStream_t to
 * launch on, or NULL to let the engine create its own.
 */
int rsx_engine_create(int device, void *stream, rsx_engine **out);
/* GPUs this process can see (0 when there is none or the runtime fails) */
int rsx_device_count(void);
/*
 * An engine over `ndev` GPUs (devices[r] may repeat: every entry gets an engine,
 * stream and workspace of its own).  Buffers of a batch are independent
 * (src/batch.rs:34-57, :79-100), so the host entry points (rsx_batch_compress,
 * rsx_batch_compress_flush, rsx_batch_decompress, rsx_batch_decompress_members,
 * rsx_batch_checksum) cut a batch into ndev contiguous ranges of items balanced
 * by input bytes, run every range on its GPU from a host thread of its own and
 * write sizes, status and payload straight into the caller's arrays: results
 * are those of a single-GPU engine, item by item.  Where CudaBatchCompressor::new
 * opens device 0 (src/batch_cuda.rs:23) a binding passes the visible devices.
 * The *_device entry points take pointers into one GPU's memory and refuse such
 * an engine (RSX_E_INVALID); ndev == 1 gives an ordinary engine.
 */
int rsx_engine_create_multi(const int *devices, int ndev, rsx_engine **out);
/* number of GPU engines behind this handle (1 for rsx_engine_create) */
int rsx_engine_num_devices(const rsx_engine *e);
void rsx_engine_destroy(rsx_engine *e);
/*
 * Engine options.  RSX_OPT_VERIFY_FHCRC (default 0): a gzip member whose header
 * carries a header CRC (FLG.FHCRC, RFC 1952 2.3.1) has it checked and is
 * RSX_BAD_DATA when it does not match; the reference skips the field unread
 * (src/decompress/mod.rs:1186-1188), which stays the default.
 */
#define RSX_OPT_VERIFY_FHCRC 1
int rsx_engine_set_option(rsx_engine *e, int option, int value);
/* last HIP / argument error text of this engine (valid until the next call) */
const char *rsx_engine_last_error(const rsx_engine *e);
/* the hipStream_t all work of this engine is launched on */
void *rsx_engine_stream(const rsx_engine *e);
/* blocks until everything queued by *_device calls has finished */
int rsx_engine_synchronize(rsx_engine *e);
/*
 * Duration in milliseconds of the kernels of the most recent *_device call
 * (all of them, first launch to last), measured with HIP events on the
 * engine's stream.  Synchronises.  Negative if none was run.
 */
float rsx_engine_last_kernel_ms(rsx_engine *e);
/*
 * Number of independently decoded parts of the most recent decompress call: equal to
 * its number of streams unless large streams were cut at sync-flush joints (the
 * chunk joints of src/compress/mod.rs:699-772), 0 if the call did not look for
 * cuts.  Diagnostic; synchronises.
 */
uint32_t rsx_engine_last_parts(rsx_engine *e);
/* ... and how many of its streams were decoded as a whole after all because a part of them did not stand by itself */
uint32_t rsx_engine_last_whole(rsx_engine *e);
/*
 * Number of streams of the most recent decompress call that the one-kernel decoder took behind the
 * two-kernel one: damaged streams, slots that are too small, streams cut into parts that did not
 * stand by themselves, and anything else the fast path gave up (a valid stream in a large enough
 * slot that shows up here is decoded correctly, but twice).  Diagnostic with no counterpart in the
 * reference; synchronises.
 */
uint32_t rsx_engine_last_fallbacks(rsx_engine *e);

/* Compressor::{deflate,zlib,gzip}_compress_bound, src/compress/mod.rs:2236-2246 */
uint64_t rsx_compress_bound(int format, uint64_t len);

/*
 * Replaces CudaBatchCompressor::compress_batch (src/batch_cuda.rs:43-150) and
 * the CPU loop it shadows (src/batch.rs:34-57).  Host pointers; returns after
 * results are in `out`.  level 0..12 as in src/api.rs:10-20.  Slot i should be
 * at least rsx_compress_bound(format, len_i) bytes; status[i] = 1 and
 * out_sizes[i] = 0 if the stream does not fit.
 */
int rsx_batch_compress(rsx_engine *e, int format, int level, const uint8_t *in,
                       const uint64_t *in_off, size_t n, uint8_t *out, const uint64_t *out_off,
                       uint64_t *out_sizes, int32_t *status);

/*
 * The encoder step of the streaming adapter, DeflateEncoder::flush_buffer
 * (src/stream.rs:42-199): raw DEFLATE of every item in 256 KiB chunks like
 * rsx_batch_compress, but with flush = RSX_FLUSH_SYNC no chunk carries the
 * final-block bit and every chunk ends with an empty stored block, so the
 * outputs of successive calls concatenate into one DEFLATE stream; the last
 * call uses RSX_FLUSH_FINISH (= rsx_batch_compress, format raw).  Give a slot
 * rsx_compress_bound(RSX_FORMAT_DEFLATE, len) + 5 bytes (src/stream.rs:107-109).
 */
int rsx_batch_compress_flush(rsx_engine *e, int level, int flush, const uint8_t *in,
                             const uint64_t *in_off, size_t n, uint8_t *out, const uint64_t *out_off,
                             uint64_t *out_sizes, int32_t *status);

/*
 * GPU counterpart of BatchDecompressor::decompress_batch (src/batch.rs:74-101;
 * the reference has no accelerator hook there).  Slot size i is the item's
 * max_out_sizes[i].  status[i] = DecompressResult; out_sizes[i] = bytes
 * produced on success.  For zlib / gzip the Adler-32 / CRC-32 + ISIZE trailer
 * is verified (src/decompress/mod.rs:1074-1240) and a mismatch is BadData.
 */
int rsx_batch_decompress(rsx_engine *e, int format, const uint8_t *in, const uint64_t *in_off,
                         size_t n, uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                         int32_t *status);

/*
 * gzip files with more than one member (pigz, concatenated .gz files).  The
 * reference decodes the first member of its input and reports the bytes it
 * consumed (src/decompress/mod.rs:1144-1240, the `(result, in_consumed,
 * out_produced)` triple); a caller reaches the following members by calling
 * again on the rest.  This entry point is that loop for a whole batch: every
 * round decodes the next member of every item that has one, appending to the
 * item's slot; out_sizes[i] = bytes of all members together.  Zero bytes behind
 * the last member are padding; anything else that is not a member makes the
 * item RSX_BAD_DATA; a member that fails gives the item that member's status.
 * Each member's CRC-32 and ISIZE are verified.  Every member depth costs one
 * chain of launches: an item with more than RSX_MAX_GZIP_MEMBERS members is
 * RSX_BAD_DATA.  An engine error in the middle (negative return) leaves the
 * items still in flight with RSX_INTERNAL_ERROR and out_sizes 0.
 */
#define RSX_MAX_GZIP_MEMBERS 4096
int rsx_batch_decompress_members(rsx_engine *e, const uint8_t *in, const uint64_t *in_off, size_t n,
                                 uint8_t *out, const uint64_t *out_off, uint64_t *out_sizes,
                                 int32_t *status);

/*
 * One step of an incremental raw-DEFLATE decoder: the engine's counterpart of
 * Decompressor::decompress_streaming (src/decompress/mod.rs:213-353), which
 * DeflateDecoder<R> drives through a sliding window (src/stream.rs:235-377).
 * The stream goes on at bit `in_bit` (0..7) of in[0]; `history` holds the last
 * bytes produced so far (up to 32768 of them are used: what a match can reach
 * back), new bytes go to out[0 .. out_cap).  The reference's state machine can
 * stop anywhere; this one stops BETWEEN TWO BLOCKS: it decodes every whole
 * block that `in` holds and `out` has room for and reports
 *   *in_next_bit   the first unread bit, counted from bit 0 of in[0] (the next
 *                  step starts at byte in_next_bit / 8, bit in_next_bit % 8),
 *   *out_produced  the bytes written to `out`,
 *   *status        RSX_SUCCESS: the final block has ended (in_next_bit = the
 *                  end of the stream); RSX_SHORT_INPUT: the next block is not
 *                  whole in `in` (more input; a step that consumed nothing
 *                  needs a longer `in`, and at the end of the input the stream
 *                  is truncated); RSX_INSUFFICIENT_SPACE: the next block does
 *                  not fit (call again with room; a step that produced nothing
 *                  needs a larger `out`); RSX_BAD_DATA: the next block is
 *                  damaged (the blocks in front of it have been delivered).
 * Host pointers, synchronous; one stream is the work of one workgroup, so this
 * is the bounded-memory path, not the fast one (rsx_batch_decompress decodes a
 * whole stream, cut at its chunk joints, on all of the GPU).
 */
int rsx_inflate_stream_step(rsx_engine *e, const uint8_t *in, size_t in_len, uint32_t in_bit,
                            const uint8_t *history, size_t history_len, uint8_t *out, size_t out_cap,
                            uint64_t *in_next_bit, size_t *out_produced, int32_t *status);

/*
 * crc32 / adler32 (src/crc32/mod.rs:332, src/adler32/mod.rs:115) over every
 * item of a batch: values[i] = f(init, item i), init = 0 for CRC-32 and 1 for
 * Adler-32 when starting a new checksum.  Host pointers.
 */
int rsx_batch_checksum(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                       const uint64_t *in_off, size_t n, uint32_t *values);
/* single-buffer convenience forms of the crate's public crc32()/adler32()
 * (src/lib.rs:15-17); *value receives the result */
int rsx_crc32(rsx_engine *e, uint32_t crc, const uint8_t *data, size_t len, uint32_t *value);
int rsx_adler32(rsx_engine *e, uint32_t adler, const uint8_t *data, size_t len, uint32_t *value);

/*
 * Device-resident variants: every pointer is device memory on the engine's
 * GPU, the work is queued on the engine's stream and the call returns without
 * waiting (use rsx_engine_synchronize or stream ordering).  They replace the
 * H2D / launch / D2H sequence of src/batch_cuda.rs:81-141 for callers whose
 * data already lives in HBM.  total_in = in_off[n] - in_off[0], max_item (the
 * largest item's input length) and, for decompress, total_out / max_out (sum
 * and maximum of the output slot sizes) are host-known scalars used to size
 * the launch and the workspace.  Items and output slots must be smaller than
 * 4 GiB - 64 KiB (larger ones are refused with RSX_E_INVALID).  A max_item
 * smaller than the longest item costs compression, not safety: such items are
 * written as stored blocks (or report InsufficientSpace).
 */
int rsx_batch_compress_device(rsx_engine *e, int format, int level, const uint8_t *in,
                              const uint64_t *in_off, size_t n, uint64_t total_in,
                              uint64_t max_item, uint8_t *out, const uint64_t *out_off,
                              uint64_t *out_sizes, int32_t *status);
int rsx_batch_decompress_device(rsx_engine *e, int format, const uint8_t *in,
                                const uint64_t *in_off, size_t n, uint64_t total_in,
                                uint64_t total_out, uint64_t max_out, uint8_t *out,
                                const uint64_t *out_off, uint64_t *out_sizes, int32_t *status);
int rsx_batch_checksum_device(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                              const uint64_t *in_off, size_t n, uint64_t total_in,
                              uint64_t max_item, uint32_t *values);

/*
 * rsx_batch_decompress_device for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  It has no counterpart in the reference.  Same arguments and
 * results as rsx_batch_decompress_device; what that call takes from the
 * engine's grow-only workspace (end-of-stream positions, checksum partials,
 * trailer sums) lives here in a 1 MiB array that is part of the library's
 * device code, one copy per GPU, so the call neither allocates nor frees.
 * Limits: n <= RSX_NOALLOC_MAX_ITEMS and total_out <= RSX_NOALLOC_MAX_TOTAL_OUT,
 * beyond them the call returns RSX_E_INVALID and does nothing.  All engines on
 * one GPU share that GPU's array: two such calls that overlap in time on one
 * GPU (different engines or streams) race; calls on one stream are safe.
 * rsx_engine_last_kernel_ms is not meaningful for a call that was captured
 * into a graph (its events are recorded when the graph runs, if at all).
 */
#define RSX_NOALLOC_MAX_ITEMS 8192
#define RSX_NOALLOC_MAX_TOTAL_OUT (1ull << 30)
int rsx_batch_decompress_device_noalloc(rsx_engine *e, int format, const uint8_t *in,
                                        const uint64_t *in_off, size_t n, uint64_t total_in,
                                        uint64_t total_out, uint64_t max_out, uint8_t *out,
                                        const uint64_t *out_off, uint64_t *out_sizes,
                                        int32_t *status);

#ifdef __cplusplus
}
#endif
#endif
