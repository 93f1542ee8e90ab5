"""not gpu: the K-Means / SOG restatements in oracle/ against fixtures produced by the REFERENCE'S OWN CODE
(oracle/make_golden_kmeans.py: gpu_ops._kmeans_taichi through oracle/taichi_shim.py, the sklearn front door under
np.random.seed, SogFormat.write decoded from its bundle).  This is what lifts "parity unpinned" from K-Means."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from oracle import datasets, kmeans as okm, refload


@pytest.fixture(scope="module")
def kref():
    with open(os.path.join(GOLDEN_DIR, "kmeans_ref.json")) as f:
        return json.load(f), np.load(os.path.join(GOLDEN_DIR, "kmeans_ref.npz"))


LLOYD = ["lloyd_1d_k32", "lloyd_9d_k48", "lloyd_24d_k32", "lloyd_45d_k64", "lloyd_empty_clusters",
         "lloyd_ties_lattice", "lloyd_3d_k1024", "lloyd_1d_k256_tails"]


@pytest.mark.parametrize("name", LLOYD)
def test_c_restatement_reproduces_the_reference_kernels_bit_for_bit(kref, name):
    cases, arr = kref
    case = cases["lloyd"][name]
    data = datasets.km_data(case["data"])
    cent, labels, _ = okm.lloyd(data, arr[name + "__init"], case["max_iter"], accumulate="f32seq")
    np.testing.assert_array_equal(labels, arr[name + "__labels"])
    np.testing.assert_array_equal(cent.view(np.uint32), arr[name + "__cent"].view(np.uint32))
    # the order-insensitive variant (f64 sums) the GPU is compared with stays within SURVEY 8(c)'s tolerance of it
    c64, l64, _ = okm.lloyd(data, arr[name + "__init"], case["max_iter"], accumulate="f64")
    assert (l64 != labels).mean() <= 2e-3
    assert abs(okm.inertia(data, c64, l64) - case["inertia"]) <= 1e-4 * case["inertia"]


def test_fixture_semantics_are_the_ones_the_survey_lists(kref):
    cases, arr = kref
    # empty cluster -> centroid left at 0 (gpu_ops.py:78-96): 80 coinciding initial centroids never win (strict <)
    assert cases["lloyd"]["lloyd_empty_clusters"]["zero_centroids"] == 80
    cent = arr["lloyd_empty_clusters__cent"]
    assert (np.abs(cent).sum(1) == 0).sum() == 80
    # exact ties on the integer lattice: the lowest centroid index wins (strict < at gpu_ops.py:68)
    data = datasets.km_data(cases["lloyd"]["lloyd_ties_lattice"]["data"])
    init = arr["lloyd_ties_lattice__init"]
    first = okm.assign(data, init)
    d = ((data[:, None, :].astype(np.float64) - init[None].astype(np.float64)) ** 2).sum(-1)
    ties = (np.sort(d, 1)[:, 0] == np.sort(d, 1)[:, 1])
    assert ties.any(), "the fixture is supposed to contain exact ties"
    np.testing.assert_array_equal(first, d.argmin(1))  # argmin returns the first minimum
    # k >= N shortcut, gpu_ops.py:30-31
    assert cases["front_door"]["k_ge_n"]["centroid_dtype"] == "float32"


@pytest.mark.skipif(not refload.available(), reason="reference not mounted (build container only)")
def test_shim_rerun_matches_the_committed_fixture(kref):
    cases, arr = kref
    shim = refload.load_gpu_ops_with_taichi_shim()
    for name in ("lloyd_ties_lattice", "lloyd_empty_clusters"):
        case = cases["lloyd"][name]
        np.random.seed(case["np_seed"])
        cent, labels = shim._kmeans_taichi(datasets.km_data(case["data"]), case["k"], case["max_iter"])
        np.testing.assert_array_equal(labels, arr[name + "__labels"])
        np.testing.assert_array_equal(cent, arr[name + "__cent"])


@pytest.mark.parametrize("name", ["sog_20k_l2", "sog_3k_l8"])
def test_sog_call_site_contract(kref, name):
    """formats/sog.py:392-449,513-552 as the reference executed it: chunk plan, codebook quantiser."""
    cases, arr = kref
    case = cases["sog"][name]
    n, level = case["n"], case["compression_level"]
    plan = okm.sog_sh_plan(n, level)
    calls = case["kmeans_calls"]
    # two scalar codebooks: min(50000, 3N) x 1 -> 256, 20 iterations (sog.py:396-403, 438-443)
    for c in calls[:2]:
        assert (c["n"], c["d"], c["k"], c["max_iter"]) == (min(50000, 3 * n), 1, 256, 20)
    chunks = calls[2:]
    assert len(chunks) == plan["num_chunks"]
    sizes = [min(plan["chunk_size"], n - i * plan["chunk_size"]) for i in range(plan["num_chunks"])]
    assert [c["n"] for c in chunks] == sizes
    assert all(c["d"] == 45 and c["max_iter"] == 10 for c in chunks)
    assert [c["k"] for c in chunks] == [min(s, plan["k_per_chunk"]) for s in sizes]
    assert case["meta"]["shN"]["count"] == sum(c["centroids_shape"][0] for c in chunks)
    # quantize_to_codebook (sog.py:408-419) against the decoded textures
    data = datasets.sog_scene(n, case["scene_seed"])
    order = np.lexsort((data["z"], data["y"], data["x"]))
    ds = data[order]
    for tex, cb_key, cols in (("scales", "scales", ["scale_0", "scale_1", "scale_2"]), ("sh0", "sh0", ["f_dc_0", "f_dc_1", "f_dc_2"])):
        cb = np.array(case["meta"][cb_key]["codebook"], dtype=np.float32)  # sog.py:403-407: np.array(sorted(f32 centroids)) is float32
        assert len(cb) == 256 and np.all(np.diff(cb) >= 0)
        # pillow's lossless WebP does not keep the colour of fully transparent pixels (exact=False): where the
        # opacity byte (sh0 alpha, sog.py:457-459) is 0 the reference's own bundle has lost the indices
        visible = arr[name + "__" + tex][:, 3] != 0
        assert visible.mean() > 0.99
        for ch, col in enumerate(cols):
            got = okm.quantize_to_codebook(ds[col], cb)
            np.testing.assert_array_equal(got[visible], arr[name + "__" + tex][visible, ch])


@pytest.mark.parametrize("name", ["sog_20k_l2", "sog_3k_l8"])
def test_sog_numeric_core_restatement_matches_the_reference_bundle(kref, name):
    """oracle/sog.py (sog.py:264-386,457-459 restated) against the textures the reference wrote"""
    import hashlib
    from oracle import sog as osog
    cases, arr = kref
    case = cases["sog"][name]
    data = datasets.sog_scene(case["n"], case["scene_seed"])
    order = osog.order(data)
    assert hashlib.sha256(order.tobytes()).digest() == arr[name + "__order_sha"].tobytes()
    ds = data[order]
    lo, hi, mins, maxs = osog.positions(ds)
    np.testing.assert_array_equal(lo, arr[name + "__means_l"][:, :3])
    np.testing.assert_array_equal(hi, arr[name + "__means_u"][:, :3])
    assert [float(m) for m in mins] == case["meta"]["means"]["mins"] and [float(m) for m in maxs] == case["meta"]["means"]["maxs"]
    np.testing.assert_array_equal(osog.quats(ds), arr[name + "__quats"])
    np.testing.assert_array_equal(osog.opacity_u8(ds), arr[name + "__sh0"][:, 3])
