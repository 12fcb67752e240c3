import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libds2i_hip.so must exist (built in-tree by __graft_entry__.build()); build it if missing."""
    import ds2i_amd
    if not os.path.exists(ds2i_amd.library_path()):
        from ds2i_amd import build as b
        b.build()
    return ds2i_amd.lib()
