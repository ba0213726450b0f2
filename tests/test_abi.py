"""The C-ABI library loads and exports every symbol include/rsx_deflate.h declares
(no compute calls: this runs without a GPU), and the host mirror validates its
arguments like src/api.rs."""
import ctypes
import os
import re

import pytest

import rsx_pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rsx_deflate.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    rsx = rsx_pkg.load()
    lib = ctypes.CDLL(rsx.library_path())
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n
    from libdeflate_rsx_amd import engine
    assert sorted(s[0] for s in engine.SYMBOLS) == names


def test_compress_bound_matches_reference_formula():
    rsx = rsx_pkg.load()
    # src/compress/mod.rs:2236-2246
    for n in (0, 1, 65535, 65536, 1 << 20, 4 << 20):
        d = n + (n // 65535 + 1) * 5 + 10
        assert rsx.compress_bound(rsx.FORMAT_DEFLATE, n) == d
        assert rsx.compress_bound(rsx.FORMAT_ZLIB, n) == d + 6
        assert rsx.compress_bound(rsx.FORMAT_GZIP, n) == d + 18
    assert rsx.compress_bound(0, 65536) == 65556 and rsx.compress_bound(0, 1 << 20) == 1048671


def test_host_mirror_argument_checks_need_no_gpu():
    rsx = rsx_pkg.load()
    with pytest.raises(ValueError, match="Compression level must be between 0 and 12"):
        rsx.Compressor(13)
    with pytest.raises(ValueError, match="Compression level must be between 0 and 12"):
        rsx.Compressor(-1)
    d = rsx.Decompressor()
    with pytest.raises(ValueError, match="exceeds safety limit"):
        d.decompress_gzip(b"\x00" * 10, 10 * 2000 + 4097)  # src/api.rs:216-229
    d.set_limit_ratio(10**9)
    d.set_max_memory_limit(1000)
    with pytest.raises(ValueError, match="exceeds maximum memory limit"):
        d.decompress_gzip(b"\x00" * 10, 1001)               # src/api.rs:231-239
    assert rsx.BatchCompressor(6).compress_batch([]) == []   # tests/batch_test.rs:41-50
    assert rsx.BatchDecompressor().decompress_batch([], []) == []


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rsx = rsx_pkg.load()
    with pytest.raises(rsx.RsxError):
        rsx.Engine(0)
    with pytest.raises(rsx.RsxError):
        rsx.BatchCompressor(6).compress_batch([b"abc"])
