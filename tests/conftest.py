import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the Map-walk extension is host code (gcc, under a second); build it if this checkout has not yet
    import __graft_entry__
    __graft_entry__.build_mapwalk()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
