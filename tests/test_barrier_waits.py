"""Static check of the generated gfx950 code: no workgroup barrier may be reachable with one of the
wave's own LDS accesses still outstanding (a ds_* instruction with no s_waitcnt lgkmcnt(0) behind
it on some path).  Runs on the CPU: hipcc -S cross-compiles the device code."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "libdeflate-rsx_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

sys.path.insert(0, os.path.join(ROOT, "tools"))


def listing(tmp_path, name, extra=()):
    out = str(tmp_path / (name + ".s"))
    src = name + (".cpp" if name == "api" else ".hip")   # (api.cpp holds the packing kernels of the host path)
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", *extra, "-x", "hip", "-S",
                    "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def hits(lines):
    import barrier_waits
    total, bad = 0, []
    for name, a, b in barrier_waits.kernels(lines):
        nb, found = barrier_waits.analyse(lines, a, b)
        total += nb
        bad += [(name, line) for line, _ in found]
    return total, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("name", ["inflate", "deflate", "checksum", "api"])
def test_no_barrier_with_an_lds_access_outstanding(tmp_path, name):
    total, bad = hits(listing(tmp_path, name))
    assert total > 0          # the listing was understood
    assert bad == []


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_the_check_sees_the_bare_barrier_of_a_plain_syncthreads_build(tmp_path):
    # Without the explicit wait the compiler leaves the pass loop's back edge unprotected; if a
    # later compiler closes that gap this test fails and WG_SYNC can become __syncthreads().
    total, bad = hits(listing(tmp_path, "inflate", ["-DRSX_PLAIN_SYNC=1"]))
    assert total > 0
    assert len(bad) >= 1 and all("inflate" in k for k, _ in bad)
