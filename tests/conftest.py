import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# One test drives the engine on device buffers owned by torch.  torch ships its own HIP
# runtime, and a process must not end up with two of them: load torch's before the engine
# library pulls in the system one.
try:
    import torch  # noqa: F401
except Exception:  # torch is optional for everything else
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def kat():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")) as f:
        return json.load(f)
