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


@pytest.mark.parametrize("name,resident", [("sog_20k_l2", True), ("sog_20k_l2", False), ("sog_3k_l8", True)])
def test_write_sog_on_the_device_reproduces_the_reference_bundle(gsx, kref, tmp_path, name, resident):
    """formats/sog_writer.py:write_sog with the real library: the textures that do not depend on the random K-Means init
    are byte-identical to the bundle the reference wrote from the same table (committed fixture); the clustered ones are
    consistent with the codebooks stored next to them"""
    import importlib
    import io
    import zipfile
    from PIL import Image
    from oracle import kmeans as okm
    cases, arr = kref
    case = cases["sog"][name]
    n = case["n"]
    data = datasets.sog_scene(n, case["scene_seed"])
    w = importlib.import_module("3dgsconverter_amd.formats.sog_writer")
    path = str(tmp_path / "out.sog")
    np.random.seed(case["np_seed"])
    # resident: the table stays in HBM for the whole core (round 6, formats/sog_device.py); else one upload / download per stage
    w.write_sog(data, path, compression_level=case["compression_level"], device_resident=resident)
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


# ---- round 6: the writer's core on a device-resident table (formats/sog_device.py, csrc/sog_table.hip) -------------------

def _sog_writer():
    import importlib
    return importlib.import_module("3dgsconverter_amd.formats.sog_writer")


def _check_core_against_numpy(core, data, level):
    """every texture of a device-resident encode against the reference's numpy expressions (oracle/sog.py restates
    formats/sog.py:264-386,457-459 line by line) on the lexsorted table; the clustered textures against their own codebooks"""
    from oracle import kmeans as okm
    n = len(data)
    tex = core["textures"]
    texels = core["width"] * core["height"]
    ds = data[osog.order(data)]
    with np.errstate(all="ignore"):
        lo, hi, mins, maxs = osog.positions(ds)
        want_q, want_a = osog.quats(ds), osog.opacity_u8(ds)
    for c in range(3):
        assert np.float32(core["mins"][c]).tobytes() == np.float32(mins[c]).tobytes()
        assert np.float32(core["maxs"][c]).tobytes() == np.float32(maxs[c]).tobytes()
    np.testing.assert_array_equal(tex["means_l"][:n, :3], lo)
    np.testing.assert_array_equal(tex["means_u"][:n, :3], hi)
    np.testing.assert_array_equal(tex["quats"][:n], want_q)
    np.testing.assert_array_equal(tex["sh0"][:n, 3], want_a)
    # fill values of the reference's np.full / np.zeros images (:300-301, :340, :425, :451, :598-606)
    for name, fill, alpha in (("means_l", 255, 255), ("means_u", 255, 255), ("quats", 255, None), ("scales", 0, 255), ("sh0", 0, None)):
        assert tex[name].shape == (texels, 4) and np.all(tex[name][n:] == fill), name
        if alpha is not None:
            assert np.all(tex[name][:n, 3] == alpha), name
    for name, cols, cbk in (("scales", ["scale_0", "scale_1", "scale_2"], "scale_codebook"), ("sh0", ["f_dc_0", "f_dc_1", "f_dc_2"], "color_codebook")):
        cb = np.asarray(core[cbk], dtype=np.float32)
        assert len(cb) == 256 and np.all(np.diff(cb) >= 0)
        for ch, col in enumerate(cols):
            np.testing.assert_array_equal(okm.quantize_to_codebook(ds[col], cb), tex[name][:n, ch])
    if core["bands"] > 0:
        coeffs = [0, 9, 24, 45][core["bands"]]
        plan = okm.sog_sh_plan(n, level)
        lab = tex["shN_labels"][:n, 0].astype(np.int64) + 256 * tex["shN_labels"][:n, 1].astype(np.int64)
        assert np.all(tex["shN_labels"][:n, 2] == 0) and np.all(tex["shN_labels"][:n, 3] == 255) and np.all(tex["shN_labels"][n:] == 0)
        chunk = np.arange(n) // plan["chunk_size"]
        assert np.all(lab // plan["k_per_chunk"] == chunk)          # labels of a chunk stay inside the chunk's slice of the palette
        assert core["palette"] == plan["k_per_chunk"] * (-(-n // plan["chunk_size"]))
        assert len(core["shn_centroid_index"]) == core["palette"] * coeffs and len(core["shn_codebook"]) == 256
    return ds


@pytest.mark.parametrize("n,level,seed", [(20000, 2, 11), (3000, 8, 12), (1100, 0, 13), (300001, 5, 14)])
def test_device_resident_core_is_numpys_bytes(gsx, n, level, seed):
    w = _sog_writer()
    data = datasets.sog_scene(n, seed)
    np.random.seed(seed)
    core = w.encode(data, level, device_resident=True)
    _check_core_against_numpy(core, data, level)
    assert core["bands"] == 3


def test_device_resident_core_ties_duplicates_and_signed_zeros(gsx):
    """the lexsort must be numpy's: long runs of ties on the primary and secondary keys, +-0.0 tie, duplicate points keep
    their table order (stable) -- visible in the quaternion texture, whose rows differ between the duplicates"""
    w = _sog_writer()
    rng = np.random.default_rng(21)
    n = 150001
    data = datasets.sog_scene(n, 21)
    for a, s in zip("xyz", (3, 2, 1.5)):
        c = np.round(rng.standard_normal(n) * s).astype(np.float32) * np.float32(0.5)
        c[rng.random(n) < 0.05] = np.float32(-0.0)
        data[a] = c
    core = w.encode(data, 2, device_resident=True)
    _check_core_against_numpy(core, data, 2)


def test_device_resident_core_equals_the_host_staged_core(gsx):
    """3 000 splats: 9 000 scalars per codebook fit, no sub-sample drawn -> both cores fit the same data with the same
    deterministic solver: every texture but the palette's (random initial centroids) must agree byte for byte"""
    w = _sog_writer()
    data = datasets.sog_scene(3000, 31)
    np.random.seed(5)
    a = w.encode(data, 8, device_resident=True)
    np.random.seed(5)
    b = w.encode(data, 8, device_resident=False)
    for name in ("means_l", "means_u", "quats", "scales", "sh0"):
        np.testing.assert_array_equal(a["textures"][name], b["textures"][name], err_msg=name)
    np.testing.assert_array_equal(np.asarray(a["scale_codebook"], np.float32), np.asarray(b["scale_codebook"], np.float32))
    np.testing.assert_array_equal(np.asarray(a["color_codebook"], np.float32), np.asarray(b["color_codebook"], np.float32))
    assert a["bands"] == b["bands"] == 3 and a["palette"] == b["palette"]
    assert [np.float32(v).tobytes() for v in a["mins"] + a["maxs"]] == [np.float32(v).tobytes() for v in b["mins"] + b["maxs"]]


def test_device_resident_core_band_detection_and_odd_rows(gsx):
    """sog.py:461-493: trailing all-zero coefficients downgrade the bands (a -0.0 counts as zero); a table widened by u1 colour
    fields (251-byte rows: data_processor.py:262-274) is packed by one host pass and takes the same path"""
    w = _sog_writer()
    from oracle import kmeans as okm
    n = 5000
    data = datasets.sog_scene(n, 41)
    for i in range(24, 45):
        data["f_rest_%d" % i] = 0.0
    data["f_rest_30"][::7] = np.float32(-0.0)
    core = w.encode(data, 9, device_resident=True)
    assert core["bands"] == 2 and len(core["shn_centroid_index"]) == core["palette"] * 24
    _check_core_against_numpy(core, data, 9)
    for i in range(9, 24):
        data["f_rest_%d" % i] = 0.0
    assert w.encode(data, 9, device_resident=True)["bands"] == 1
    for i in range(9):
        data["f_rest_%d" % i] = 0.0
    core0 = w.encode(data, 9, device_resident=True)
    assert core0["bands"] == 0 and "shN_labels" not in core0["textures"]
    # no f_rest fields at all
    d0 = datasets.sog_scene(n, 42, sh_degree=0)
    c0 = w.encode(d0, 0, device_resident=True)
    assert c0["bands"] == 0
    _check_core_against_numpy(c0, d0, 0)
    # odd row size
    wide = np.zeros(n, dtype=np.dtype(datasets.sog_scene(8, 1).dtype.descr + [("red", "u1"), ("green", "u1"), ("blue", "u1")]))
    src = datasets.sog_scene(n, 43)
    for nm in src.dtype.names:
        wide[nm] = src[nm]
    assert wide.dtype.itemsize % 4 != 0
    np.random.seed(3)
    cw = w.encode(wide, 2, device_resident=True)
    np.random.seed(3)
    cs = w.encode(src, 2, device_resident=True)
    for name in cs["textures"]:
        np.testing.assert_array_equal(cw["textures"][name], cs["textures"][name], err_msg=name)


def test_device_resident_core_declines_what_it_does_not_take(gsx):
    w = _sog_writer()
    sd = __import__("importlib").import_module("3dgsconverter_amd.formats.sog_device")
    small = datasets.sog_scene(500, 1)
    with pytest.raises(sd.NotEligible):
        w.encode(small, 0, device_resident=True)
    assert w.encode(small, 0)["n"] == 500                        # ... and the default falls back to the host-staged core
    bad = datasets.sog_scene(4000, 2)
    bad["y"][17] = np.nan
    with pytest.raises(sd.NotEligible):
        w.encode(bad, 0, device_resident=True)
    f8 = datasets.sog_scene(4000, 3).astype([(nm, "f8" if nm == "opacity" else "f4") for nm in datasets.sog_scene(8, 1).dtype.names])
    with pytest.raises(sd.NotEligible):
        w.encode(f8, 0, device_resident=True)


def test_device_resident_core_degenerate_axes_and_wide_ranges(gsx):
    """a two-valued axis (every value is an extreme: more candidates than the list holds -> the reference's expression on the
    host column), values spanning 24 orders of magnitude, a one-sided axis"""
    w = _sog_writer()
    rng = np.random.default_rng(51)
    n = 120000
    data = datasets.sog_scene(n, 51)
    data["x"] = np.where(rng.random(n) < 0.5, np.float32(1.25), np.float32(-0.75)).astype(np.float32)
    data["y"] = (rng.standard_normal(n) * np.exp(rng.uniform(-12, 12, n))).astype(np.float32)
    data["z"] = (np.abs(rng.standard_normal(n)) + 2.0).astype(np.float32)
    with np.errstate(all="ignore"):
        core = w.encode(data, 4, device_resident=True)
        _check_core_against_numpy(core, data, 4)
    st = core["stats"]
    assert st["uncertain_alpha"] <= 0.005 * n and st["uncertain_positions"] <= 0.2 * 3 * n, st


def test_device_resident_core_takes_any_field_layout(gsx):
    """fields in another order with foreign fields in between (the device reads each float32 field at its own offset), and rows
    longer than the 512 bytes the device tile holds (packed by one host pass): the same images as the standard table"""
    w = _sog_writer()
    src = datasets.sog_scene(20011, 71)
    names = list(src.dtype.names)
    rng = np.random.default_rng(71)
    shuffled = [names[i] for i in rng.permutation(len(names))]
    descr = []
    for i, nm in enumerate(shuffled):
        descr.append((nm, "f4"))
        if i % 7 == 3:
            descr.append(("pad_%d" % i, "i4"))
    mixed = np.zeros(len(src), dtype=descr)
    long_rows = np.zeros(len(src), dtype=descr + [("blob", "f4", (80,))])
    assert mixed.dtype.itemsize % 4 == 0 and mixed.dtype.itemsize <= 512 < long_rows.dtype.itemsize
    for nm in names:
        mixed[nm] = src[nm]
        long_rows[nm] = src[nm]
    outs = []
    for tab in (src, mixed, long_rows):
        np.random.seed(9)
        outs.append(w.encode(tab, 3, device_resident=True))
    for other in outs[1:]:
        assert other["bands"] == outs[0]["bands"] and other["palette"] == outs[0]["palette"]
        for name in outs[0]["textures"]:
            np.testing.assert_array_equal(other["textures"][name], outs[0]["textures"][name], err_msg=name)
        np.testing.assert_array_equal(other["shn_centroid_index"], outs[0]["shn_centroid_index"])


def _sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("name", ["sog_120k_l5", "sog_60k_l0_bands2", "sog_40k_l9_lattice", "sog_30k_l3_bands1"])
def test_device_resident_core_against_reference_run_hashes(gsx, name):
    """tests/golden/sog_ref_hashes.json: what the UN-PATCHED reference's SogFormat.write put into its bundle for these tables
    (oracle/make_golden_sog.py, run in the build container) -- a larger scene, band downgrades to 2 and 1, lattice coordinates with
    ties and signed zeros.  Every init-independent texture (padding included), the means' min / max, the detected bands and the
    palette size of the device-resident core must be the reference's."""
    from oracle import make_golden_sog as mgs
    with open(os.path.join(GOLDEN_DIR, "sog_ref_hashes.json")) as f:
        case = json.load(f)[name]
    data = mgs.build(case)
    w = _sog_writer()
    np.random.seed(case["np_seed"])
    core = w.encode(data, case["level"], device_resident=True)
    tex = core["textures"]
    assert len(tex["means_l"]) == case["texels"]
    assert _sha16(tex["means_l"]) == case["sha"]["means_l"] and _sha16(tex["means_u"]) == case["sha"]["means_u"]
    assert _sha16(tex["quats"]) == case["sha"]["quats"]
    assert _sha16(tex["sh0"][:, 3]) == case["sha"]["sh0_alpha"] and _sha16(tex["scales"][:, 3]) == case["sha"]["scales_alpha"]
    assert [float(m) for m in core["mins"]] == case["means"]["mins"] and [float(m) for m in core["maxs"]] == case["means"]["maxs"]
    assert core["bands"] == case["bands"] and core.get("palette", 0) == case["palette"]
    if case["sha"]["labels_pad_alpha"] is not None:
        assert _sha16(tex["shN_labels"][case["n"]:, 3]) == case["sha"]["labels_pad_alpha"]
        assert not tex["shN_labels"][case["n"]:].any()      # np.zeros in the reference (sog.py:598); see the fixture's note on alpha 0


def test_device_resident_core_on_a_memory_mapped_table(gsx, tmp_path):
    """a table that is a view of a mapped file (np.memmap, copy-on-write and read-only) is uploaded by the plain copy and gives
    the textures of the same table in anonymous memory"""
    w = _sog_writer()
    data = datasets.sog_scene(40000, 77)
    path = tmp_path / "table.bin"
    data.tofile(path)
    np.random.seed(3)
    want = w.encode(data, 8, device_resident=True)
    for mode in ("c", "r"):
        mapped = np.memmap(path, dtype=data.dtype, mode=mode)
        assert gsx._lib.file_backed(mapped)
        np.random.seed(3)
        got = w.encode(mapped, 8, device_resident=True)
        for name in want["textures"]:
            np.testing.assert_array_equal(got["textures"][name], want["textures"][name], err_msg="%s %s" % (mode, name))
        del mapped


def test_device_resident_core_on_a_scene_far_from_the_origin(gsx):
    """every coordinate large against its axis' range (a capture that is not centred: x in [10, 30], y in [200, 201], z one-sided):
    the +-5 ulp bracket of numpy's float32 log covers a sizeable part of a texel step, a fifth and more of the position texels
    are listed as uncertain -- far beyond the first list's capacity.  The core lists them all on a second pass (it used to hand
    the table to the host-staged path) and stays numpy's bytes"""
    w = _sog_writer()
    n = 60000
    rng = np.random.default_rng(91)
    data = datasets.sog_scene(n, 91)
    data["x"] = rng.uniform(10, 30, n).astype(np.float32)
    data["y"] = rng.uniform(200, 201, n).astype(np.float32)
    data["z"] = (np.abs(rng.standard_normal(n)) + 4).astype(np.float32)
    core = w.encode(data, 4, device_resident=True)
    assert core["stats"]["uncertain_positions"] > n // 8 + 4096, core["stats"]
    _check_core_against_numpy(core, data, 4)
    # opacities by the thousand outside the range where the sigmoid's byte can be certified: NaN, +-inf, +-1e30
    data["opacity"][rng.integers(0, n, 20000)] = np.float32(np.nan)
    data["opacity"][rng.integers(0, n, 5000)] = np.float32(np.inf)
    data["opacity"][rng.integers(0, n, 5000)] = np.float32(-1e30)
    with np.errstate(all="ignore"):
        core = w.encode(data, 4, device_resident=True)
        _check_core_against_numpy(core, data, 4)


@pytest.mark.parametrize("front,back,deg", [(0, 3, 3), (1, 0, 3), (2, 1, 2), (3, 2, 1), (0, 1, 0), (5, 6, 3), (1, 251, 3), (3, 248, 3),
                                            (0, 264, 3)])      # ... 500- and 499-byte rows (the largest off the grid), 512 on it
def test_device_resident_core_reads_rows_off_the_four_byte_grid(gsx, front, back, deg):
    """round 6: rows whose size is not a multiple of 4 (the reference's converter appends three u1 colour fields before it calls the
    SOG writer, converter.py:243-252 -> 251 bytes) and float fields at odd offsets (u1 fields in front) are read on the device as
    they are -- a field assembled from two words of the LDS tile -- instead of being packed by a host pass.  Same textures as the
    same values in a plain 4-byte table; sizes around the 128-row tiles; band detection and the key extremes included"""
    w = _sog_writer()
    sd = __import__("importlib").import_module("3dgsconverter_amd.formats.sog_device")
    for n in (1100, 128 * 40 + 1, 30011):
        src = datasets.sog_scene(n, 50 + n % 7, sh_degree=deg)
        if deg >= 2:
            for i in range(3 * ((deg + 1) ** 2 - 1) - 4, 3 * ((deg + 1) ** 2 - 1)):
                src["f_rest_%d" % i][: n - 1] = 0          # the only non-zero of these fields in the last row
        descr = [("pre%d" % i, "u1") for i in range(front)] + src.dtype.descr + [("post%d" % i, "u1") for i in range(back)]
        odd = np.zeros(n, dtype=np.dtype(descr))
        rng = np.random.default_rng(n)
        for nm in odd.dtype.names:
            odd[nm] = src[nm] if nm in src.dtype.names else rng.integers(0, 256, n, dtype=np.uint8)
        rows, lay = sd.table_layout(odd)
        assert rows is odd and lay.row_bytes == odd.dtype.itemsize      # no host pack
        np.random.seed(9)
        a = w.encode(odd, 7, device_resident=True)
        np.random.seed(9)
        b = w.encode(src, 7, device_resident=True)
        assert a["bands"] == b["bands"] == deg
        for name in b["textures"]:
            np.testing.assert_array_equal(a["textures"][name], b["textures"][name], err_msg="%s n=%d" % (name, n))
        assert [np.float32(v).tobytes() for v in a["mins"] + a["maxs"]] == [np.float32(v).tobytes() for v in b["mins"] + b["maxs"]]
    _check_core_against_numpy(a, odd, 7)
