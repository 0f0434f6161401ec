import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def lpamd_pkg():
    """The product package (its directory name has a hyphen, so no plain `import`)."""
    return importlib.import_module("linear-programming_amd")


@pytest.fixture(scope="session")
def lpamd():
    return lpamd_pkg()


GOLDEN = os.path.join(ROOT, "tests", "golden")
