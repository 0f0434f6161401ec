import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """Fatal signals name their origin (a fault of this thread with its address, or a signal sent by the kernel or
    another process) and the native frames before Python's faulthandler prints the Python ones: a crash "inside pure
    Python" right after native calls cannot be told apart otherwise (tests/native/segv_info.c)."""
    import ctypes
    helper = os.path.join(ROOT, "tests", "native", "libsegvinfo.so")
    if os.path.exists(helper):
        ctypes.CDLL(helper).segv_info_install()


def lpamd_pkg():
    """The product package (its directory name has a hyphen, so no plain `import`)."""
    return importlib.import_module("linear-programming_amd")


@pytest.fixture(scope="session")
def lpamd():
    return lpamd_pkg()


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _device_index_checks(request):
    """With a library built by `make dbg-checks` (LPAMD_LIBRARY): no kernel index check may have failed during the test."""
    yield
    if "LPAMD_LIBRARY" not in os.environ or "gpu" not in request.keywords:
        return
    import ctypes
    lib = lpamd_pkg().library()
    fn = getattr(lib, "lpamd_dbg_check_failures", None)
    if fn is None:
        return
    out = (ctypes.c_uint32 * 32)()
    assert fn(out) == 0
    assert not any(out), "kernel index checks failed: %s" % {i: v for i, v in enumerate(out) if v}
