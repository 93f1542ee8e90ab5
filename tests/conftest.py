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
    # the suite asserts bit-exact masks: a numpy whose float32 reductions are ordered differently must FAIL here, not warn
    os.environ.setdefault("GSX_STRICT_NUMPY", "1")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build the HIP library (hipcc
    cross-compiles without a GPU) and the oracle's C restatement once, like __graft_entry__.build()."""
    import shutil
    lib = os.path.join(ROOT, "3dgsconverter_amd", "libgsx_hip.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        sys.path.insert(0, os.path.join(ROOT, "3dgsconverter_amd"))
        try:
            import build as _build  # 3dgsconverter_amd/build.py
            _build.build()
        finally:
            sys.path.pop(0)
            sys.modules.pop("build", None)


@pytest.fixture(scope="session")
def golden_cases():
    with open(os.path.join(GOLDEN_DIR, "cases.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def large_cases():
    """hash-only fixtures at BASELINE.json's full sizes (oracle/make_golden_large.py, run against the reference)"""
    with open(os.path.join(GOLDEN_DIR, "large_cases.json")) as f:
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
