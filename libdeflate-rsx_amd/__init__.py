"""libdeflate-rsx_amd -- MI355X-native batch DEFLATE engine.

Python view of librsx_deflate.so (the C ABI in include/rsx_deflate.h) with the
same surface as the reference crate's hot path:

    batch.BatchCompressor / batch.BatchDecompressor   (src/batch.rs)
    api.Compressor / api.Decompressor                  (src/api.rs)
    crc32 / adler32                                    (src/lib.rs:15-17)

There is no CPU fallback in this package: every call runs the HIP kernels and
raises if the shared library or a GPU is missing.
"""
from .engine import (Engine, RsxError, FORMAT_DEFLATE, FORMAT_ZLIB, FORMAT_GZIP, SUCCESS,
                     BAD_DATA, SHORT_OUTPUT, INSUFFICIENT_SPACE, SHORT_INPUT, CHECKSUM_CRC32,
                     CHECKSUM_ADLER32, FLUSH_SYNC, FLUSH_FINISH, OPT_VERIFY_FHCRC, compress_bound, default_engine, device_count,
                     library_path, load_library)
from .batch import BatchCompressor, BatchDecompressor
from .api import Compressor, Decompressor, crc32, adler32

__all__ = [
    "Engine", "RsxError", "BatchCompressor", "BatchDecompressor", "Compressor", "Decompressor",
    "crc32", "adler32", "compress_bound", "default_engine", "device_count", "library_path", "load_library", "OPT_VERIFY_FHCRC",
    "FORMAT_DEFLATE", "FORMAT_ZLIB", "FORMAT_GZIP", "SUCCESS", "BAD_DATA", "SHORT_OUTPUT",
    "INSUFFICIENT_SPACE", "SHORT_INPUT", "CHECKSUM_CRC32", "CHECKSUM_ADLER32", "FLUSH_SYNC", "FLUSH_FINISH",
]
