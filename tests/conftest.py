import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    arrays = np.load(os.path.join(GOLDEN, "golden.npz"), allow_pickle=False)
    scal = json.load(open(os.path.join(GOLDEN, "golden.json")))
    issue42 = np.fromfile(os.path.join(GOLDEN, "issue42_stream.bin"), dtype=np.uint8)
    return arrays, scal, issue42


@pytest.fixture(scope="session")
def oracle():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The compiled reference, or None where oracle/_ref was not built."""
    from oracle import Reference
    try:
        return Reference()
    except FileNotFoundError:
        return None


if os.environ.get("SVB_ABORT_TRACE"):
    # diagnostic: native backtrace on SIGABRT / SIGSEGV (tools/native/abort_trace.c)
    import ctypes
    import subprocess
    import tempfile
    _so = os.path.join(tempfile.gettempdir(), "svb_abort_trace.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", _so,
                           os.path.join(ROOT, "tools", "native", "abort_trace.c")])
    ctypes.CDLL(_so).abort_trace_install()
