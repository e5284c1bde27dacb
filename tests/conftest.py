import os
import sys

import pytest

# exercise the halo-tile conv kernel on the small test shapes too (the production gate needs >= 128 workgroups)
os.environ.setdefault("GG_HALO_MIN_BLOCKS", "1")
os.environ.setdefault("GG_HALO_MIN_BLOCKS_2D", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
