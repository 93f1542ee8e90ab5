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


def test_write_sog_on_the_device_reproduces_the_reference_bundle(gsx, kref, tmp_path):
    """formats/sog_writer.py:write_sog with the real library: the textures that do not depend on the random K-Means init
    are byte-identical to the bundle the reference wrote from the same table (committed fixture); the clustered ones are
    consistent with the codebooks stored next to them"""
    import importlib
    import io
    import zipfile
    from PIL import Image
    from oracle import kmeans as okm
    cases, arr = kref
    name = "sog_20k_l2"
    case = cases["sog"][name]
    n = case["n"]
    data = datasets.sog_scene(n, case["scene_seed"])
    w = importlib.import_module("3dgsconverter_amd.formats.sog_writer")
    path = str(tmp_path / "out.sog")
    np.random.seed(case["np_seed"])
    w.write_sog(data, path, compression_level=case["compression_level"])
    with zipfile.ZipFile(path) as zf:
        meta = json.loads(zf.read("meta.json"))
        tex = {f[:-5]: np.asarray(Image.open(io.BytesIO(zf.read(f))).convert("RGBA"), dtype=np.uint8).reshape(-1, 4)
               for f in zf.namelist() if f.endswith(".webp")}
    for t in ("means_l", "means_u", "quats"):
        np.testing.assert_array_equal(tex[t][:n], arr[name + "__" + t])
    np.testing.assert_array_equal(tex["sh0"][:n, 3], arr[name + "__sh0"][:, 3])
    assert meta["means"] == case["meta"]["means"] and meta["count"] == n
    assert meta["shN"]["count"] == case["meta"]["shN"]["count"] and meta["shN"]["bands"] == 3
    assert tex["shN_centroids"].shape == arr[name + "__shN_centroids"].shape
    ds = data[np.lexsort((data["z"], data["y"], data["x"]))]
    for t, cols in (("scales", ["scale_0", "scale_1", "scale_2"]), ("sh0", ["f_dc_0", "f_dc_1", "f_dc_2"])):
        cb = np.array(meta[t]["codebook"], dtype=np.float32)
        vis = tex[t][:n, 3] != 0
        for ch, col in enumerate(cols):
            np.testing.assert_array_equal(okm.quantize_to_codebook(ds[col], cb)[vis], tex[t][:n, ch][vis])
    lab = tex["shN_labels"][:n, 0].astype(np.int64) + 256 * tex["shN_labels"][:n, 1].astype(np.int64)
    assert lab.max() < meta["shN"]["count"]


def _np_positions(v):
    t = np.sign(v) * np.log(np.abs(v) + 1.0)                       # formats/sog.py:280-281
    mn, mx = np.min(t), np.max(t)
    return np.clip((t - mn) / (mx - mn) * 65535, 0, 65535).astype(np.uint16), mn, mx     # :293-295


@pytest.mark.parametrize("kind", ["scene", "wide", "tiny", "mixed", "grid", "one_sided"])
def test_log_positions_device_path_is_numpys_bytes(gsx, kind):
    """gsx_sog_positions (float64 log + rounding certificate, numpy only for flagged texels) == numpy's own float32
    expression of formats/sog.py:279-309, texel for texel, min / max bits included"""
    lib = gsx._lib
    rng = np.random.default_rng(5)
    n = 1_000_003
    v = {"scene": lambda: rng.standard_normal(n) * 3.0,
         "wide": lambda: rng.standard_normal(n) * np.exp(rng.uniform(-12, 12, n)),
         "tiny": lambda: rng.standard_normal(n) * 1e-6,
         "mixed": lambda: np.concatenate([rng.standard_normal(n - 6) * 50.0, [0.0, -0.0, 1e-30, -1e-30, 3.0e5, -2.0e5]]),
         "grid": lambda: np.round(rng.standard_normal(n) * 40.0) * 0.25,
         "one_sided": lambda: np.abs(rng.standard_normal(n)) + 2.0}[kind]().astype(np.float32)
    stats = {}
    got, mn, mx = lib.sog_positions(v, stats)
    want, wmn, wmx = _np_positions(v)
    assert np.float32(mn).tobytes() == np.float32(wmn).tobytes() and np.float32(mx).tobytes() == np.float32(wmx).tobytes()
    np.testing.assert_array_equal(got, want)
    # the device decides nearly all of it: a texel is undecided when numpy's possible results (+-5 ulp of the log) straddle a
    # u16 step, i.e. with probability ~ 10 ulp(|l|) / (range / 65535)
    t = np.sign(v) * np.log(np.abs(v) + 1.0)
    expected = float(np.mean(10 * np.spacing(np.abs(t)) / ((wmx - wmn) / 65535.0)))
    assert stats["uncertain"] <= n * (1.5 * expected + 1e-3), (stats, expected)


def test_log_positions_on_the_reference_bundle(gsx, kref):
    cases, arr = kref
    for name in ("sog_20k_l2", "sog_3k_l8"):
        case = cases["sog"][name]
        data = datasets.sog_scene(case["n"], case["scene_seed"])
        ds = data[np.lexsort((data["z"], data["y"], data["x"]))]
        for c, a in enumerate("xyz"):
            u, mn, mx = gsx._lib.sog_positions(ds[a])
            np.testing.assert_array_equal((u & 0xff).astype(np.uint8), arr[name + "__means_l"][:case["n"], c])
            np.testing.assert_array_equal((u >> 8).astype(np.uint8), arr[name + "__means_u"][:case["n"], c])
            assert float(mn) == case["meta"]["means"]["mins"][c] and float(mx) == case["meta"]["means"]["maxs"][c]
        np.testing.assert_array_equal(gsx._lib.sog_alpha(ds["opacity"]), arr[name + "__sh0"][:case["n"], 3])


def test_sigmoid_alpha_device_path_is_numpys_bytes(gsx):
    rng = np.random.default_rng(6)
    n = 2_000_001
    o = (rng.standard_normal(n) * 4.0).astype(np.float32)
    o[:8] = [0.0, -0.0, 90.0, -90.0, 17.0, -17.0, 1e-8, -1e-8]
    # logits of k/255 exactly: texel boundaries
    kk = np.arange(1, 255, dtype=np.float64) / 255.0
    o[8:8 + 254] = np.log(kk / (1 - kk)).astype(np.float32)
    stats = {}
    got = gsx._lib.sog_alpha(o, stats)
    with np.errstate(all="ignore"):
        want = np.clip(1.0 / (1.0 + np.exp(-o)) * 255, 0, 255).astype(np.uint8)
    np.testing.assert_array_equal(got, want)
    assert stats["uncertain"] <= 0.005 * n, stats
