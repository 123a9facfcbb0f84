import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def liw():
    """The product package (its name is not a Python identifier)."""
    return importlib.import_module("2dliw-slam_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("2dliw-slam_amd.synth")


@pytest.fixture(scope="session")
def pyoracle():
    from oracle import pyoracle as po
    po.build()
    return po
