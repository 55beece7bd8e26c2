import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): oracle/oracle.py over oracle/gpe_oracle.c."""
    mod = importlib.import_module("oracle.oracle")
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def gpe():
    """The product package (directory name carries a hyphen, hence importlib)."""
    return importlib.import_module("gpu-physics-engine_amd")
