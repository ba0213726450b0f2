// engine.h -- internal state of an rsx_engine and the launch helpers shared by
// the kernel translation units.  Not part of the public ABI (include/rsx_deflate.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rsx_deflate.h"

struct HostPipe;   // api.cpp: streams, pinned buffers and copy threads of the host entry points

struct rsx_engine {
    // An engine made by rsx_engine_create_multi for more than one device owns one engine per device and no GPU
    // state of its own: its host entry points cut a batch into one contiguous range of items per device.
    std::vector<rsx_engine *> subs;
    HostPipe *pipe = nullptr;   // created by the first host entry point call
    bool verify_fhcrc = false;  // RSX_OPT_VERIFY_FHCRC

    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;
    std::string err;

    // grow-only device workspace (ws), carved per call by ws_take() from the arena the call
    // selected: the workspace (ws_reserve) or this device's static scratch (ws_use_static)
    uint8_t *ws = nullptr;
    size_t ws_size = 0;
    uint8_t *arena = nullptr;
    size_t arena_size = 0;
    size_t ws_used = 0;

    // device address of the static scratch array of inflate.hip on this engine's GPU
    // (hipGetSymbolAddress at engine creation); shared by every engine on that GPU
    uint8_t *static_scratch = nullptr;
    size_t static_scratch_size = 0;

    int num_cus = 0;           // multiprocessors of this engine's GPU
    bool inflate_stats = false;  // RSX_INFLATE_STATS was set when the engine was created
    bool inflate_split = true;   // two-kernel decoder (RSX_INFLATE_FUSED=1 at engine creation turns it off)
    bool inflate_parts = true;   // few large streams are cut at sync-flush markers (RSX_INFLATE_WHOLE=1 turns it off)

    // constant tables in device memory (built once in rsx_engine_create)
    uint32_t *d_crc_tables = nullptr;  // see checksum.hip for the layout

    const uint32_t *d_last_fallbacks = nullptr;   // device word: streams of the last decompress call the one-kernel decoder took
    const uint32_t *d_last_parts = nullptr;   // device word: parts of the last decompress call (in the workspace)
    // The host entry points run a batch slice by slice: they add every slice's words up here (device memory,
    // 4 words: parts, streams taken out behind the first / the second kernel, streams decoded twice)
    uint32_t *d_diag = nullptr;

    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    bool timed = false;
};

#define RSX_HIP(e, call)                                                              \
    do {                                                                              \
        hipError_t _st = (call);                                                      \
        if (_st != hipSuccess) {                                                      \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_st);            \
            return RSX_E_DEVICE;                                                      \
        }                                                                             \
    } while (0)

// Ensures the workspace holds at least `bytes`, makes it the arena and resets the carve pointer.
int ws_reserve(rsx_engine *e, size_t bytes);
// Makes the device's static scratch the arena (no allocation); RSX_E_INVALID if `bytes` exceed it.
int ws_use_static(rsx_engine *e, size_t bytes);
// Carves `bytes` (256-byte aligned) out of the arena; nullptr if exhausted.
void *ws_take(rsx_engine *e, size_t bytes);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- kernel launchers (each defined next to its kernels) ---------------------

// Workgroup barrier.  The LDS writes of a wave must have landed before it signals the
// barrier.  __syncthreads() alone leaves that to the compiler's wait-count insertion.  With
// hipcc 7.2.0 -O3 the barrier at the head of the speculative decoder's pass loop (inflate.hip,
// `for (pass ...) { WG_SYNC(); ... lane_exit[i] = ...; }`) comes out as a bare s_barrier on the
// back edge although the loop body ends in ds_write instructions, and a faster wave then reads
// the previous pass's lane_exit.  The wait is therefore written out; everywhere else it
// duplicates one the compiler already emits.  tools/wg_sync_isa.sh compares both builds and
// tools/barrier_waits.py lists every barrier reachable with an LDS access outstanding.
#ifndef RSX_PLAIN_SYNC
#define WG_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); } while (0)
#else   // only for comparing the generated code
#define WG_SYNC() __syncthreads()
#endif

// checksum.hip
int checksum_tables_create(rsx_engine *e);
size_t checksum_ws_bytes(size_t n, uint64_t total_in);
// items are in[starts[i] .. starts[i] + len_i) with len_i = lens ? lens[i] : starts[i+1]-starts[i];
// total_in bounds the sum of lengths, max_item the largest one.  trusted_max: max_item was derived from
// the offsets by this library (host entry points); a caller's word is not used to lay out the partials
int checksum_launch(rsx_engine *e, int kind, uint32_t init, const uint8_t *in,
                    const uint64_t *starts, const uint64_t *lens, size_t n, uint64_t total_in,
                    uint64_t max_item, uint32_t *values, bool trusted_max);

// inflate.hip
// The `format` of inflate_launch may carry this flag: a gzip header's FHCRC field is checked (RFC 1952 2.3.1;
// the reference skips the field, src/decompress/mod.rs:1186-1188) and a mismatch is RSX_BAD_DATA.
constexpr int kFormatMask = 0xFF;
constexpr int kFormatVerifyFhcrc = 0x100;
// looks up this GPU's copy of the static scratch array (call with the engine's device current)
int inflate_static_scratch(rsx_engine *e);
// split: the two-kernel decoder (tokens in the workspace) with the one-kernel decoder for what it leaves;
// otherwise the one-kernel decoder alone (small workspace)
size_t inflate_ws_bytes(size_t n, uint64_t total_in, uint64_t total_out, bool split);
// total_out / max_out: sum and maximum of the output slot sizes
// Items that are not the slices of two prefix arrays (all device pointers): item i is
// in[in_off[i], in_hi[i]) -> out[out_off[i], out_hi[i]); *in_base and total_in bound the bytes of
// `in` the kernels may touch; in_used (may be null, zlib / gzip only) receives the bytes of the item
// its stream took, header and trailer included, 0 for a failed item.
struct InflateSlices {
    const uint64_t *in_hi, *out_hi, *in_base;
    uint64_t *in_used;
};
// slices == nullptr: the usual layout, item i = [in_off[i], in_off[i + 1]) -> [out_off[i], out_off[i + 1])
int inflate_launch(rsx_engine *e, int format, const uint8_t *in, const uint64_t *in_off, size_t n,
                   uint64_t total_in, uint64_t total_out, uint64_t max_out, uint8_t *out,
                   const uint64_t *out_off, uint64_t *out_sizes, int32_t *status, bool split,
                   const InflateSlices *slices = nullptr);

// one resumable item through the one-kernel decoder; ctl: ten u64 of device memory, see inflate.hip
int inflate_resume_launch(rsx_engine *e, const uint8_t *d_in, uint64_t in_len, uint8_t *d_out, uint64_t *ctl);

// deflate.hip
size_t deflate_ws_bytes(size_t n, uint64_t total_in, uint64_t max_item, int level);
int deflate_launch(rsx_engine *e, int format, int level, const uint8_t *in, const uint64_t *in_off,
                   size_t n, uint64_t total_in, uint64_t max_item, uint8_t *out,
                   const uint64_t *out_off, uint64_t *out_sizes, int32_t *status, int flush,
                   bool trusted_max);
