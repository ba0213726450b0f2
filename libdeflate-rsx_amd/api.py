"""Mirror of src/api.rs (Compressor / Decompressor) and the crate-level
crc32 / adler32 (src/lib.rs:15-17) over the GPU engine.  Errors are raised as
ValueError (io::ErrorKind::InvalidInput), RuntimeError (ErrorKind::Other) or
InvalidDataError (ErrorKind::InvalidData) with the reference's messages."""
from .engine import (CHECKSUM_ADLER32, CHECKSUM_CRC32, FORMAT_DEFLATE, FORMAT_GZIP, FORMAT_ZLIB,
                     compress_bound, default_engine)


class InvalidDataError(RuntimeError):
    pass


def crc32(crc, data, engine=None):
    return int((engine or default_engine()).batch_checksum([data], CHECKSUM_CRC32, crc)[0])


def adler32(adler, data, engine=None):
    return int((engine or default_engine()).batch_checksum([data], CHECKSUM_ADLER32, adler)[0])


class Compressor:
    """src/api.rs:5-133"""

    def __init__(self, level, engine=None):
        if not 0 <= level <= 12:
            raise ValueError("Compression level must be between 0 and 12")  # api.rs:11-16
        self.level = level
        self._engine = engine

    def _run(self, data, fmt, cap=None):
        eng = self._engine or default_engine()
        res, _ = eng.batch_compress([data], fmt, self.level, None if cap is None else [cap])
        return res[0]

    def _vec(self, data, fmt):
        r = self._run(data, fmt)
        if r is None:
            raise RuntimeError("Insufficient space")  # api.rs:96
        return r

    def compress_deflate(self, data):
        return self._vec(data, FORMAT_DEFLATE)

    def compress_zlib(self, data):
        return self._vec(data, FORMAT_ZLIB)

    def compress_gzip(self, data):
        return self._vec(data, FORMAT_GZIP)

    def _into(self, data, output, fmt, msg):
        r = self._run(data, fmt, len(output))
        if r is None:
            raise RuntimeError(msg)
        output[:len(r)] = r
        return len(r)

    def compress_deflate_into(self, data, output):
        return self._into(data, output, FORMAT_DEFLATE, "Insufficient space")  # api.rs:31

    def compress_zlib_into(self, data, output):
        return self._into(data, output, FORMAT_ZLIB, "Compression failed")  # api.rs:43

    def compress_gzip_into(self, data, output):
        return self._into(data, output, FORMAT_GZIP, "Compression failed")  # api.rs:54

    def deflate_compress_bound(self, size):
        return compress_bound(FORMAT_DEFLATE, size)

    def zlib_compress_bound(self, size):
        return compress_bound(FORMAT_ZLIB, size)

    def gzip_compress_bound(self, size):
        return compress_bound(FORMAT_GZIP, size)


class Decompressor:
    """src/api.rs:135-301: expected-size API with the zip-bomb guards of :216-239."""

    def __init__(self, engine=None):
        self.max_memory_limit = (1 << 64) - 1
        self.limit_ratio = 2000
        self._engine = engine

    def set_max_memory_limit(self, limit):
        self.max_memory_limit = limit

    def set_limit_ratio(self, ratio):
        self.limit_ratio = ratio

    def _helper(self, data, expected_size, fmt):
        limit = len(data) * self.limit_ratio + 4096
        if expected_size > limit:
            raise ValueError("Expected size %d exceeds safety limit for input size %d"
                             % (expected_size, len(data)))
        if expected_size > self.max_memory_limit:
            raise ValueError("Expected size %d exceeds maximum memory limit %d"
                             % (expected_size, self.max_memory_limit))
        return self._run(data, expected_size, fmt)

    def _run(self, data, cap, fmt):
        eng = self._engine or default_engine()
        res, _ = eng.batch_decompress([data], [cap], fmt)
        if res[0] is None:
            raise InvalidDataError("Decompression failed")  # api.rs:258-261
        return res[0]

    def decompress_deflate(self, data, expected_size):
        return self._helper(data, expected_size, FORMAT_DEFLATE)

    def decompress_zlib(self, data, expected_size):
        return self._helper(data, expected_size, FORMAT_ZLIB)

    def decompress_gzip(self, data, expected_size):
        return self._helper(data, expected_size, FORMAT_GZIP)

    def _into(self, data, output, fmt):
        # the _into variants skip the limits (tests/security_limit.rs:274-297)
        r = self._run(data, len(output), fmt)
        output[:len(r)] = r
        return len(r)

    def decompress_deflate_into(self, data, output):
        return self._into(data, output, FORMAT_DEFLATE)

    def decompress_zlib_into(self, data, output):
        return self._into(data, output, FORMAT_ZLIB)

    def decompress_gzip_into(self, data, output):
        return self._into(data, output, FORMAT_GZIP)
