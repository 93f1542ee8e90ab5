"""-m gpu: the SOG writer's numeric core on the device (csrc/sog.hip) against the textures of bundles the reference's own
SogFormat.write produced (tests/golden/kmeans_ref.npz) and against numpy on adversarial inputs.  Bar: byte-exact."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import datasets, sog as osog

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kref():
    with open(os.path.join(GOLDEN_DIR, "kmeans_ref.json")) as f:
        return json.load(f), np.load(os.path.join(GOLDEN_DIR, "kmeans_ref.npz"))


@pytest.mark.parametrize("name", ["sog_20k_l2", "sog_3k_l8"])
def test_lexsort_and_quats_against_the_reference_bundle(gsx, kref, name):
    lib = gsx._lib
    lib.require_hip()
    cases, arr = kref
    case = cases["sog"][name]
    data = datasets.sog_scene(case["n"], case["scene_seed"])
    order = lib.lexsort3(data["z"], data["y"], data["x"])      # formats/sog.py:264
    assert order.dtype == np.int64
    assert hashlib.sha256(order.tobytes()).digest() == arr[name + "__order_sha"].tobytes()
    ds = data[order]
    q = np.column_stack([ds["rot_0"], ds["rot_1"], ds["rot_2"], ds["rot_3"]])
    np.testing.assert_array_equal(lib.sog_quats(q), arr[name + "__quats"])


def test_lexsort_ties_signed_zeros_and_size(gsx):
    lib = gsx._lib
    rng = np.random.default_rng(1)
    for n in (1, 2, 1000, 300001, 2_000_000):
        # few distinct values per key: long runs of ties on the primary and secondary keys; +-0.0 must tie
        k = [np.round(rng.standard_normal(n) * s).astype(np.float32) * np.float32(0.5) for s in (3, 2, 1.5)]
        for c in k:
            c[rng.random(n) < 0.05] = np.float32(-0.0)
        np.testing.assert_array_equal(lib.lexsort3(k[0], k[1], k[2]), np.lexsort((k[0], k[1], k[2])))
    x = (rng.standard_normal(500000) * 1e3).astype(np.float32)
    np.testing.assert_array_equal(lib.lexsort3(x, x[::-1].copy(), -x), np.lexsort((x, x[::-1], -x)))


def test_quats_random_and_axis_aligned(gsx):
    lib = gsx._lib
    rng = np.random.default_rng(2)
    q = rng.standard_normal((400003, 4)).astype(np.float32)
    q[:4] = np.eye(4, dtype=np.float32)               # exactly one component
    q[4:8] = -np.eye(4, dtype=np.float32)
    q[8] = [0.5, 0.5, 0.5, 0.5]                       # four-way tie of the maximum: first index
    q[9] = [-0.5, 0.5, -0.5, 0.5]
    q[10:20] *= np.float32(1e-20)                     # tiny norms
    q[20:30] *= np.float32(1e15)
    ds = np.zeros(len(q), dtype=[("rot_0", "f4"), ("rot_1", "f4"), ("rot_2", "f4"), ("rot_3", "f4")])
    for c in range(4):
        ds["rot_%d" % c] = q[:, c]
    with np.errstate(all="ignore"):
        want = osog.quats(ds)
    np.testing.assert_array_equal(lib.sog_quats(q), want)
