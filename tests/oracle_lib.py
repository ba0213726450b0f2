"""ctypes view of the CPU restatement (oracle/build/librsx_oracle.so).

Test infrastructure: imported only by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "build", "librsx_oracle.so")

SUCCESS, BAD_DATA, SHORT_OUTPUT, INSUFFICIENT_SPACE, SHORT_INPUT = range(5)
FMT_DEFLATE, FMT_ZLIB, FMT_GZIP = 0, 1, 2
FLUSH_NONE, FLUSH_SYNC, FLUSH_FINISH = 0, 1, 2

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u8p, u64p, i32p = C.c_void_p, C.c_void_p, C.c_void_p
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_uint32, u8p, C.c_size_t]
        L.orc_adler32.restype = C.c_uint32
        L.orc_adler32.argtypes = [C.c_uint32, u8p, C.c_size_t]
        L.orc_compress_bound.restype = C.c_size_t
        L.orc_compress_bound.argtypes = [C.c_int, C.c_size_t]
        L.orc_decompress.restype = C.c_int
        L.orc_decompress.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_compressor_new.restype = C.c_void_p
        L.orc_compressor_new.argtypes = [C.c_int]
        L.orc_compressor_free.argtypes = [C.c_void_p]
        L.orc_compress.restype = C.c_int
        L.orc_compress.argtypes = [C.c_void_p, C.c_int, u8p, C.c_size_t, u8p, C.c_size_t,
                                   C.POINTER(C.c_size_t)]
        L.orc_compress_raw.restype = C.c_int
        L.orc_compress_raw.argtypes = [C.c_void_p, u8p, C.c_size_t, u8p, C.c_size_t, C.c_int,
                                       C.POINTER(C.c_size_t)]
        L.orc_hc_find_match_kat.argtypes = [u8p, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                            C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_make_huffman_code.argtypes = [C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_batch_compress.argtypes = [C.c_int, C.c_int, u8p, u64p, C.c_size_t, u8p, u64p,
                                         u64p, i32p, C.c_int]
        L.orc_batch_decompress.argtypes = [C.c_int, u8p, u64p, C.c_size_t, u8p, u64p, u64p,
                                           i32p, C.c_int]
        _lib = L
    return _lib


def _buf(b):
    """bytes-like -> (keepalive ndarray, pointer)"""
    a = np.frombuffer(b, dtype=np.uint8) if not isinstance(b, np.ndarray) else b
    if a.size == 0:
        a = np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data


def crc32(data, crc=0):
    a, p = _buf(data)
    return lib().orc_crc32(crc, p, len(data))


def adler32(data, adler=1):
    a, p = _buf(data)
    return lib().orc_adler32(adler, p, len(data))


def compress_bound(fmt, n):
    return lib().orc_compress_bound(fmt, n)


def decompress(data, out_cap, fmt=FMT_DEFLATE):
    """-> (result code, in_consumed, output bytes)"""
    a, p = _buf(data)
    out = np.empty(max(out_cap, 1), dtype=np.uint8)
    ic, op = C.c_size_t(0), C.c_size_t(0)
    r = lib().orc_decompress(fmt, p, len(data), out.ctypes.data, out_cap, C.byref(ic), C.byref(op))
    return r, ic.value, out[:op.value].tobytes()


def compress(data, level=6, fmt=FMT_DEFLATE, out_cap=None):
    """-> compressed bytes, or None on InsufficientSpace"""
    L = lib()
    c = L.orc_compressor_new(level)
    if not c:
        raise ValueError("compression level %d is outside 0..12" % level)
    try:
        a, p = _buf(data)
        cap = compress_bound(fmt, len(data)) if out_cap is None else out_cap
        out = np.empty(max(cap, 1), dtype=np.uint8)
        n = C.c_size_t(0)
        r = L.orc_compress(c, fmt, p, len(data), out.ctypes.data, cap, C.byref(n))
        return None if r else out[:n.value].tobytes()
    finally:
        L.orc_compressor_free(c)


def compress_raw(data, level, flush):
    L = lib()
    c = L.orc_compressor_new(level)
    try:
        a, p = _buf(data)
        cap = compress_bound(FMT_DEFLATE, len(data)) + 16
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        r = L.orc_compress_raw(c, p, len(data), out.ctypes.data, cap, flush, C.byref(n))
        return None if r else out[:n.value].tobytes()
    finally:
        L.orc_compressor_free(c)


def hc_find_match(data, pos, max_depth, nice_len):
    a, p = _buf(data)
    l, o = C.c_int(0), C.c_int(0)
    lib().orc_hc_find_match_kat(p, len(data), pos, max_depth, nice_len, C.byref(l), C.byref(o))
    return l.value, o.value


def batch_compress(flat, in_off, fmt, level, threads=8):
    """flat: uint8 ndarray, in_off: uint64[n+1] -> (out, out_off, sizes, status)"""
    n = len(in_off) - 1
    lens = np.diff(in_off.astype(np.int64))
    bounds = np.array([compress_bound(fmt, int(x)) for x in lens], dtype=np.uint64)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(bounds, out=out_off[1:])
    out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    flat = np.ascontiguousarray(flat)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    lib().orc_batch_compress(fmt, level, flat.ctypes.data if flat.size else None,
                             in_off.ctypes.data, n, out.ctypes.data, out_off.ctypes.data,
                             sizes.ctypes.data, status.ctypes.data, threads)
    return out, out_off, sizes, status


def batch_decompress(flat, in_off, out_off, fmt, threads=8):
    n = len(in_off) - 1
    out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
    sizes = np.zeros(n, dtype=np.uint64)
    status = np.zeros(n, dtype=np.int32)
    flat = np.ascontiguousarray(flat)
    in_off = np.ascontiguousarray(in_off, dtype=np.uint64)
    out_off = np.ascontiguousarray(out_off, dtype=np.uint64)
    lib().orc_batch_decompress(fmt, flat.ctypes.data if flat.size else None, in_off.ctypes.data,
                               n, out.ctypes.data, out_off.ctypes.data, sizes.ctypes.data,
                               status.ctypes.data, threads)
    return out, sizes, status


def make_huffman_code(freqs, max_len):
    """huffman_comp.rs:125-155 -> (code lengths, bit-reversed codewords) for the given symbol frequencies."""
    f = np.ascontiguousarray(np.asarray(freqs, dtype=np.uint32))
    lens = np.zeros(len(f), dtype=np.uint8)
    codes = np.zeros(len(f), dtype=np.uint32)
    lib().orc_make_huffman_code(len(f), max_len, f.ctypes.data, lens.ctypes.data, codes.ctypes.data)
    return lens, codes
