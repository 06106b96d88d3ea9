import os
import sys

import pytest
import torch  # noqa: F401  (first: see homulator_amd/__init__.py on HIP-runtime load order)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_small():
    from oracle.homoracle import Oracle
    return Oracle(10, 6, 2)


@pytest.fixture(scope="session")
def oracle_mid():
    from oracle.homoracle import Oracle
    return Oracle(12, 8, 3)
