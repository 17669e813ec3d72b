import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    p = entry.load_package()
    if not os.path.exists(p.LIB_PATH):
        p.build()
    return p


@pytest.fixture(scope="session")
def scenes(pkg):
    import importlib
    return importlib.import_module(entry.PKG_NAME + ".scenes")


@pytest.fixture(scope="session")
def oracle():
    return entry.load_oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def dragon(scenes, golden_dir):
    return scenes.load_crtscene(os.path.join(golden_dir, "dragon.crtscene"))
