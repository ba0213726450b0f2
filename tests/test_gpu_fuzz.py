"""A short, seeded run of tools/fuzz_gpu.py inside the GPU suite: damaged and intact streams of every
zlib strategy against the oracle, and compressor output of random levels through zlib."""
import importlib.util
import os

import pytest

import rsx_pkg

pytestmark = pytest.mark.gpu


def _tool():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "fuzz_gpu.py")
    spec = importlib.util.spec_from_file_location("fuzz_gpu", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_inflate_fuzz_seeds():
    rsx = rsx_pkg.load()
    _tool().fuzz_inflate(rsx.default_engine(), 60.0, 5000, max_seeds=12)   # exits non-zero on a mismatch


def test_compress_fuzz_seeds():
    rsx = rsx_pkg.load()
    _tool().fuzz_compress(rsx.default_engine(), rsx, 60.0, 9000, max_seeds=40)
