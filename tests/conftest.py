import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def halo_hint(monkeypatch):
    """Run small shapes on the halo-tile conv kernel (gg_conv_desc.path_hint = 1 lifts its >= 128-workgroup production gate);
    every test WITHOUT this fixture runs under the production dispatch."""
    from jointimagegeneration_amd import ops
    monkeypatch.setattr(ops, "PATH_HINT", 1)
    yield

