"""ctypes binding of include/rsx_deflate.h."""
import ctypes as C
import os
import threading

import numpy as np

FORMAT_DEFLATE, FORMAT_ZLIB, FORMAT_GZIP = 0, 1, 2
SUCCESS, BAD_DATA, SHORT_OUTPUT, INSUFFICIENT_SPACE, SHORT_INPUT = range(5)
INTERNAL_ERROR = 100   # the engine made no progress on an item (never a property of the data)
CHECKSUM_CRC32, CHECKSUM_ADLER32 = 0, 1
FLUSH_SYNC, FLUSH_FINISH = 1, 2   # FlushMode, src/compress/mod.rs:424-429

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LOCK = threading.Lock()

# every symbol include/rsx_deflate.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("rsx_engine_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("rsx_device_count", C.c_int, []),
    ("rsx_engine_create_multi", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(_P)]),
    ("rsx_engine_num_devices", C.c_int, [_P]),
    ("rsx_engine_destroy", None, [_P]),
    ("rsx_engine_set_option", C.c_int, [_P, C.c_int, C.c_int]),
    ("rsx_engine_last_error", C.c_char_p, [_P]),
    ("rsx_engine_stream", _P, [_P]),
    ("rsx_engine_synchronize", C.c_int, [_P]),
    ("rsx_engine_last_kernel_ms", C.c_float, [_P]),
    ("rsx_engine_last_parts", C.c_uint32, [_P]),
    ("rsx_engine_last_whole", C.c_uint32, [_P]),
    ("rsx_engine_last_fallbacks", C.c_uint32, [_P]),
    ("rsx_compress_bound", C.c_uint64, [C.c_int, C.c_uint64]),
    ("rsx_batch_compress", C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    ("rsx_batch_compress_flush", C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    ("rsx_batch_decompress", C.c_int, [_P, C.c_int, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    ("rsx_batch_decompress_members", C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, _P, _P]),
    ("rsx_inflate_stream_step", C.c_int, [_P, _P, C.c_size_t, C.c_uint32, _P, C.c_size_t, _P, C.c_size_t,
                                          C.POINTER(C.c_uint64), C.POINTER(C.c_size_t), C.POINTER(C.c_int32)]),
    ("rsx_batch_checksum", C.c_int, [_P, C.c_int, C.c_uint32, _P, _P, C.c_size_t, _P]),
    ("rsx_crc32", C.c_int, [_P, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("rsx_adler32", C.c_int, [_P, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_uint32)]),
    ("rsx_batch_compress_device", C.c_int,
     [_P, C.c_int, C.c_int, _P, _P, C.c_size_t, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    ("rsx_batch_decompress_device", C.c_int,
     [_P, C.c_int, _P, _P, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
    ("rsx_batch_checksum_device", C.c_int,
     [_P, C.c_int, C.c_uint32, _P, _P, C.c_size_t, C.c_uint64, C.c_uint64, _P]),
    ("rsx_batch_decompress_device_noalloc", C.c_int,
     [_P, C.c_int, _P, _P, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64, _P, _P, _P, _P]),
]
OPT_VERIFY_FHCRC = 1
MAX_GZIP_MEMBERS = 4096
NOALLOC_MAX_ITEMS = 8192
NOALLOC_MAX_TOTAL_OUT = 1 << 30


class RsxError(RuntimeError):
    """Engine-level failure (the reference's shim would fall back to the CPU here)."""


def library_path():
    # RSX_LIBRARY: another build of the same library (kernel tuning experiments)
    return os.environ.get("RSX_LIBRARY") or os.path.join(_HERE, "lib", "librsx_deflate.so")


def load_library():
    """Loads librsx_deflate.so and binds every declared symbol; raises if it is missing."""
    global _LIB
    with _LOCK:
        if _LIB is None:
            path = library_path()
            if not os.path.exists(path):
                raise RsxError("%s is missing: build it with `make -C libdeflate-rsx_amd/csrc` "
                               "(or __graft_entry__.build())" % path)
            lib = C.CDLL(path)
            for name, res, args in SYMBOLS:
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _LIB = lib
    return _LIB


def compress_bound(fmt, n):
    return load_library().rsx_compress_bound(fmt, n)


def device_count():
    """GPUs the library can see (rsx_device_count)."""
    return int(load_library().rsx_device_count())


def _np_ptr(a):
    return a.ctypes.data if a.size else None


class Engine:
    """One engine = one GPU (rsx_engine).  `stream` is a raw hipStream_t or None."""

    def __init__(self, device=0, stream=None, devices=None):
        """`devices`: a list of GPU ordinals for an engine over several GPUs (rsx_engine_create_multi);
        otherwise one engine on `device`."""
        self._lib = load_library()
        h = _P()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            r = self._lib.rsx_engine_create_multi(arr, len(devices), C.byref(h))
            if r != 0:
                raise RsxError("rsx_engine_create_multi(%r) failed with %d (no usable GPU?)" % (list(devices), r))
        else:
            r = self._lib.rsx_engine_create(device, stream, C.byref(h))
            if r != 0:
                raise RsxError("rsx_engine_create(device=%d) failed with %d (no usable GPU?)" % (device, r))
        self._h = h
        self.device = device

    @property
    def num_devices(self):
        return int(self._lib.rsx_engine_num_devices(self._h))

    def set_option(self, option, value):
        self._check(self._lib.rsx_engine_set_option(self._h, option, int(value)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rsx_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, r):
        if r != 0:
            raise RsxError("engine error %d: %s" % (r, self._lib.rsx_engine_last_error(self._h).decode()))

    @property
    def stream(self):
        return self._lib.rsx_engine_stream(self._h)

    def synchronize(self):
        self._check(self._lib.rsx_engine_synchronize(self._h))

    def last_kernel_ms(self):
        return self._lib.rsx_engine_last_kernel_ms(self._h)

    def last_parts(self):
        """Independently decoded parts of the last decompress call (0: it did not look for cuts)."""
        return int(self._lib.rsx_engine_last_parts(self._h))

    def last_whole(self):
        """Streams of the last decompress call that were decoded as a whole after a part of them failed."""
        return int(self._lib.rsx_engine_last_whole(self._h))

    def last_fallbacks(self):
        """Streams of the last decompress call that the one-kernel decoder took behind the two-kernel one."""
        return int(self._lib.rsx_engine_last_fallbacks(self._h))

    # ---- host-pointer calls (numpy) -------------------------------------------------

    @staticmethod
    def _flatten(items):
        lens = np.fromiter((len(x) for x in items), dtype=np.uint64, count=len(items))
        off = np.zeros(len(items) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        flat = np.frombuffer(b"".join(bytes(x) for x in items), dtype=np.uint8)
        return flat, off

    def batch_compress(self, items, fmt=FORMAT_DEFLATE, level=6, slot_sizes=None):
        """-> (list of bytes-or-None, status ndarray)"""
        n = len(items)
        flat, in_off = self._flatten(items)
        if slot_sizes is None:
            slot_sizes = [compress_bound(fmt, len(x)) for x in items]
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.asarray(slot_sizes, dtype=np.uint64), out=out_off[1:])
        out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
        sizes = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        self._check(self._lib.rsx_batch_compress(self._h, fmt, level, _np_ptr(flat), in_off.ctypes.data,
                                                 n, out.ctypes.data, out_off.ctypes.data,
                                                 sizes.ctypes.data, status.ctypes.data))
        res = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])].tobytes() if status[i] == 0 else None
               for i in range(n)]
        return res, status[:n]

    def batch_compress_flush(self, items, level=6, flush=FLUSH_SYNC):
        """raw DEFLATE pieces for a streaming encoder (src/stream.rs:42-199): with FLUSH_SYNC the
        pieces of successive calls concatenate into one stream -> (list of bytes-or-None, status)"""
        n = len(items)
        flat, in_off = self._flatten(items)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.asarray([compress_bound(FORMAT_DEFLATE, len(x)) + 5 for x in items], dtype=np.uint64),
                  out=out_off[1:])
        out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
        sizes = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        self._check(self._lib.rsx_batch_compress_flush(self._h, level, flush, _np_ptr(flat), in_off.ctypes.data,
                                                       n, out.ctypes.data, out_off.ctypes.data,
                                                       sizes.ctypes.data, status.ctypes.data))
        res = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])].tobytes() if status[i] == 0 else None
               for i in range(n)]
        return res, status[:n]

    def batch_decompress(self, items, max_out_sizes, fmt=FORMAT_DEFLATE):
        """-> (list of bytes-or-None, status ndarray)"""
        n = len(items)
        flat, in_off = self._flatten(items)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.asarray(max_out_sizes, dtype=np.uint64), out=out_off[1:])
        out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
        sizes = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        self._check(self._lib.rsx_batch_decompress(self._h, fmt, _np_ptr(flat), in_off.ctypes.data, n,
                                                   out.ctypes.data, out_off.ctypes.data,
                                                   sizes.ctypes.data, status.ctypes.data))
        res = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])].tobytes() if status[i] == 0 else None
               for i in range(n)]
        return res, status[:n]

    def batch_decompress_members(self, items, max_out_sizes):
        """gzip items that may hold several members -> (list of bytes-or-None, status ndarray)"""
        n = len(items)
        flat, in_off = self._flatten(items)
        out_off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(np.asarray(max_out_sizes, dtype=np.uint64), out=out_off[1:])
        out = np.empty(max(int(out_off[-1]), 1), dtype=np.uint8)
        sizes = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.int32)
        self._check(self._lib.rsx_batch_decompress_members(self._h, _np_ptr(flat), in_off.ctypes.data, n,
                                                           out.ctypes.data, out_off.ctypes.data,
                                                           sizes.ctypes.data, status.ctypes.data))
        res = [out[int(out_off[i]):int(out_off[i]) + int(sizes[i])].tobytes() if status[i] == 0 else None
               for i in range(n)]
        return res, status[:n]

    def inflate_stream_step(self, data, in_bit, history, out_cap):
        """rsx_inflate_stream_step -> (status, next_bit, bytes produced)"""
        data, history = bytes(data), bytes(history)
        out = C.create_string_buffer(max(out_cap, 1))
        nb, prod, st = C.c_uint64(0), C.c_size_t(0), C.c_int32(0)
        self._check(self._lib.rsx_inflate_stream_step(self._h, C.c_char_p(data) if data else None, len(data), in_bit,
                                                      C.c_char_p(history) if history else None, len(history),
                                                      C.cast(out, _P), out_cap, C.byref(nb), C.byref(prod), C.byref(st)))
        return st.value, nb.value, out.raw[:prod.value]

    def batch_checksum(self, items, kind, init=None):
        n = len(items)
        if init is None:
            init = 0 if kind == CHECKSUM_CRC32 else 1
        flat, in_off = self._flatten(items)
        vals = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(self._lib.rsx_batch_checksum(self._h, kind, init, _np_ptr(flat), in_off.ctypes.data,
                                                 n, vals.ctypes.data))
        return vals[:n]

    # ---- device-pointer calls (raw addresses; torch tensors via .data_ptr()) -------

    def batch_decompress_device(self, fmt, d_in, d_in_off, n, total_in, total_out, max_out, d_out,
                                d_out_off, d_sizes, d_status):
        self._check(self._lib.rsx_batch_decompress_device(self._h, fmt, d_in, d_in_off, n, total_in,
                                                          total_out, max_out, d_out, d_out_off,
                                                          d_sizes, d_status))

    def batch_decompress_device_noalloc(self, fmt, d_in, d_in_off, n, total_in, total_out, max_out, d_out,
                                        d_out_off, d_sizes, d_status):
        """rsx_batch_decompress_device without any device allocation (static scratch; see the header)"""
        self._check(self._lib.rsx_batch_decompress_device_noalloc(self._h, fmt, d_in, d_in_off, n, total_in,
                                                                  total_out, max_out, d_out, d_out_off,
                                                                  d_sizes, d_status))

    def batch_compress_device(self, fmt, level, d_in, d_in_off, n, total_in, max_item, d_out, d_out_off,
                              d_sizes, d_status):
        self._check(self._lib.rsx_batch_compress_device(self._h, fmt, level, d_in, d_in_off, n,
                                                        total_in, max_item, d_out, d_out_off, d_sizes,
                                                        d_status))

    def batch_checksum_device(self, kind, init, d_in, d_in_off, n, total_in, max_item, d_values):
        self._check(self._lib.rsx_batch_checksum_device(self._h, kind, init, d_in, d_in_off, n,
                                                        total_in, max_item, d_values))


_DEFAULT = {}
_DEFAULT_LOCK = threading.Lock()   # not _LOCK: Engine() loads the library under that one


def default_engine(device=0):
    """Process-wide engine per device, created lazily like the reference's
    OnceLock<Option<CudaBatchCompressor>> (src/batch.rs:7-8, :23-27)."""
    with _DEFAULT_LOCK:
        e = _DEFAULT.get(device)
        if e is None:
            e = Engine(device)
            _DEFAULT[device] = e
        return e
