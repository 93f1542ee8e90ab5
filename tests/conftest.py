import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLDEN_DIR, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_arrays():
    return np.load(os.path.join(GOLDEN_DIR, "arrays.npz"))


@pytest.fixture(scope="session")
def gsx():
    """The product package (its directory name starts with a digit, hence importlib)."""
    return importlib.import_module("3dgsconverter_amd")


def f32_from_hex(h: str) -> np.float32:
    return np.frombuffer(bytes.fromhex(h), dtype=np.float32)[0]


def sha16(b: bytes) -> str:
    import hashlib
    return hashlib.sha256(b).hexdigest()[:16]
