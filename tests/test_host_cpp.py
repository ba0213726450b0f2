"""The C++ host classes (libdeflate-rsx_amd/host/rsx_deflate.hpp: BatchCompressor,
BatchDecompressor, Compressor, Decompressor, crc32, adler32 as in src/batch.rs and
src/api.rs) are exercised by a C++ program that checks every result against the CPU oracle.
The binary is built by __graft_entry__.build() (tests/host_cpp/Makefile)."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "host_cpp", "test_host")


def _run(*args):
    if not os.path.exists(BIN):
        pytest.fail("tests/host_cpp/test_host is not built (run __graft_entry__.build())")
    r = subprocess.run([BIN, *args], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def test_argument_checks_need_no_gpu():
    # level range, zip-bomb guards, overlap checks and the bound formulas never reach the engine
    assert "ok" in _run("--no-gpu")


@pytest.mark.gpu
def test_host_classes_on_gpu():
    assert "host c++ tests ok" in _run()
