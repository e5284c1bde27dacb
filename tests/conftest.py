import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(params=[1, 4, 6], ids=["box512", "box256", "box1024"])
def halo_hint(monkeypatch, request):
    """Run small shapes on the halo-tile conv kernel (gg_conv_desc.path_hint = 1 / 4 lifts its >= 128-workgroup production gate;
    1: 512-position boxes, 4: the 256-position 3-D boxes of under-filled grids, 6: the 1024-position 3-D boxes of filled grids); every test WITHOUT this fixture runs under the
    production dispatch."""
    from jointimagegeneration_amd import ops
    monkeypatch.setattr(ops, "PATH_HINT", request.param)
    yield

