import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def ctx720():
    from ros2_mono_vo_amd import Context
    c = Context(max_width=1280, max_height=720, nfeatures=2000, max_points=8192)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx480():
    from ros2_mono_vo_amd import Context
    c = Context(max_width=640, max_height=480, nfeatures=1000, max_points=8192)
    yield c
    c.close()
