"""-m gpu: the HIP SOR path (through the C ABI) against the oracle / golden vectors.
Bar: mean_dists bit-identical (f32), statistics bit-identical, survivor mask bit-identical."""
import numpy as np
import pytest

from conftest import f32_from_hex, sha16
from oracle import datasets, sor as osor

pytestmark = pytest.mark.gpu

BRUTE, GRID = 1, 2

SMALL = ["sor_u100k_k8_s1", "sor_u100k_k8_s10p5", "sor_u20k_k16_s1", "sor_u50k_k32_s1",
         "sor_clustered30k_k16_s2", "sor_dups8k_k8_s1", "sor_lattice17_k6_s1", "sor_lattice17_k26_s0p5",
         "sor_tiny10_k16", "sor_tiny17_k16", "sor_u20k_int1", "sor_u20k_int5", "sor_u20k_int10",
         "sor_centered40k_k16_s1"]


@pytest.fixture(scope="module")
def lib(gsx):
    gsx._lib.require_hip()
    return gsx._lib


def _explain(md_gpu, md_ref):
    bad = np.nonzero(md_gpu.view(np.uint32) != md_ref.view(np.uint32))[0]
    if len(bad) == 0:
        return "ok"
    with np.errstate(all="ignore"):
        rel = np.abs(md_gpu[bad].astype(np.float64) - md_ref[bad]) / np.abs(md_ref[bad])
    return "%d / %d mean_dists differ; first %s gpu=%s ref=%s max rel %.3g" % (
        len(bad), len(md_ref), bad[:8], md_gpu[bad[:8]], md_ref[bad[:8]], np.nanmax(rel))


def _check_against_case(lib, name, case, arrays, algo):
    xyz = datasets.make(case["dataset"])
    assert sha16(xyz.tobytes()) == case["xyz_sha"]
    res = lib.sor_filter(xyz, case["k_used"], case["sigma_used"], algo=algo, want_info=True)
    md = res["mean_dists"]
    if sha16(md.tobytes()) != case["mean_dists_sha"]:
        ref = osor.mean_dists_ckdtree(xyz, case["k_used"])
        pytest.fail("%s algo=%d: %s (info %s)" % (name, algo, _explain(md, ref), res["info"]))
    for got, key in ((res["mean"], "mean_hex"), (res["std"], "std_hex"), (res["threshold"], "threshold_hex")):
        want = f32_from_hex(case[key])
        assert np.float32(got).tobytes() == want.tobytes() or (np.isnan(got) and np.isnan(want)), (name, key, got, want)
    np.testing.assert_array_equal(np.packbits(res["mask"]), arrays[name + "__mask"])
    assert int(res["mask"].sum()) == case["survivors"]
    return res


@pytest.mark.parametrize("algo", [BRUTE, GRID])
@pytest.mark.parametrize("name", SMALL)
def test_sor_golden_small(lib, golden_cases, golden_arrays, name, algo):
    _check_against_case(lib, name, golden_cases["sor"][name], golden_arrays, algo)


@pytest.mark.parametrize("name", ["sor_u1m_k16_s1", "sor_u1m_k16_s2"])
def test_sor_1m_kat_grid(lib, golden_cases, golden_arrays, name):
    """BASELINE.json configs[1]: 1M splats, k=16 -- 848 169 survivors at sigma 1."""
    res = _check_against_case(lib, name, golden_cases["sor"][name], golden_arrays, GRID)
    assert res["info"]["n_fallback"] < 0.1 * 1_000_000


def test_sor_1m_kat_brute(lib, golden_cases, golden_arrays):
    _check_against_case(lib, "sor_u1m_k16_s1", golden_cases["sor"]["sor_u1m_k16_s1"], golden_arrays, BRUTE)


def test_soa_and_rows_agree(lib):
    xyz = datasets.uniform(30000, 10.0, 21)
    a = lib.sor_filter(xyz, 16, 1.0, algo=GRID)
    b = lib.sor_filter((xyz[:, 0].copy(), xyz[:, 1].copy(), xyz[:, 2].copy()), 16, 1.0, algo=GRID)
    np.testing.assert_array_equal(a["mean_dists"].view(np.uint32), b["mean_dists"].view(np.uint32))
    np.testing.assert_array_equal(a["mask"], b["mask"])


@pytest.mark.parametrize("algo", [BRUTE, GRID])
@pytest.mark.parametrize("k", [1, 3, 8, 9, 17, 25, 33, 50, 64])
def test_k_buckets(lib, algo, k):
    """every list-capacity bucket (9/17/33/65) and the k<8 sequential mean"""
    xyz = datasets.clustered(6000, seed=k)
    ref = osor.mean_dists_ckdtree(xyz, k)
    res = lib.sor_filter(xyz, k, 1.5, algo=algo)
    assert _explain(res["mean_dists"], ref) == "ok"
    m, s, t = osor.threshold_numpy(ref, 1.5)
    assert np.float32(t).tobytes() == np.float32(res["threshold"]).tobytes()
    np.testing.assert_array_equal(res["mask"], ref < t)


@pytest.mark.parametrize("algo", [BRUTE, GRID])
def test_query_subrange_device_api(gsx, lib, algo):
    """sharded queries (what each rank of a multi-GPU run does): any index range of the cloud"""
    xyz = datasets.uniform(50000, 10.0, 33)
    ref = osor.mean_dists_ckdtree(xyz, 16)
    ctx = lib.Context(0)
    n = len(xyz)
    cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
    d = [ctx.alloc(4 * n).upload(c) for c in cols]
    for q0, qc in ((0, n), (12345, 20000), (n - 777, 777), (5, 1)):
        out = ctx.alloc(4 * qc)
        info = ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, q0, qc, 16, out.ptr, algo=algo, want_info=True)
        got = out.download(np.float32, qc)
        assert _explain(got, ref[q0:q0 + qc]) == "ok", (q0, qc, info)
        out.free()
    for a in d:
        a.free()
    ctx.close()


@pytest.mark.parametrize("algo,n,k,nshares", [(GRID, 200000, 16, 8), (GRID, 60000, 32, 3), (GRID, 5000, 8, 5),
                                              (BRUTE, 3000, 16, 4), (GRID, 40000, 16, 1)])
def test_query_shares_sum_to_the_whole(lib, algo, n, k, nshares):
    """multi-GPU building block: the per-rank shares (spatial slabs of bricks) are disjoint, cover every
    query, leave +0.0 elsewhere, and their plain f32 sum is bit-identical to the unsharded result"""
    xyz = datasets.uniform(n, 10.0, 77) if n != 60000 else datasets.clustered(n, 5)
    ref = osor.mean_dists_ckdtree(xyz, k)
    ctx = lib.Context(0)
    cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
    d = [ctx.alloc(4 * n).upload(c) for c in cols]
    out = ctx.alloc(4 * n)
    total = np.zeros(n, np.float32)
    owners = np.zeros(n, np.int32)
    sizes = []
    for share in range(nshares):
        out.upload(np.full(n, np.nan, np.float32))  # the call must overwrite everything
        ctx.sor_knn_share(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, k, share, nshares, out.ptr, algo=algo)
        got = out.download(np.float32, n)
        assert np.isfinite(got).all()
        mine = got != 0
        assert not np.signbit(got[~mine]).any()  # +0.0, not -0.0
        owners += mine
        sizes.append(int(mine.sum()))
        total = total + got  # what the sum all-reduce does
    zero_ref = int((ref == 0).sum())  # exact duplicates k+1 deep would have mean 0 (none in these clouds)
    assert zero_ref == 0
    assert (owners == 1).all(), (sizes, int((owners != 1).sum()))
    assert _explain(total, ref) == "ok", sizes
    if algo == GRID and n >= 200000:
        assert max(sizes) < 1.25 * n / nshares, sizes  # slabs of a uniform cloud are balanced
    for a in d + [out]:
        a.free()
    ctx.close()


@pytest.mark.parametrize("name,n,k", [("uniform", 200000, 16), ("uniform", 120000, 32), ("uniform", 50000, 8),
                                      ("clustered", 60000, 16), ("lattice", 0, 16), ("uniform", 30000, 50)])
def test_mfma_filter_variant_is_exact_too(lib, name, n, k):
    """knn_brick's optional phase-1 filter (ctx param filter_mfma=1: bf16-split v_mfma_f32_32x32x16_bf16 with a
    conservative slack) must select a superset of the true neighbours, i.e. give bit-identical results"""
    xyz = {"uniform": lambda: datasets.uniform(n, 10.0, 5), "clustered": lambda: datasets.clustered(n, 7),
           "lattice": lambda: datasets.lattice(30)}[name]()
    n = len(xyz)
    ref = osor.mean_dists_ckdtree(xyz, k)
    ctx = lib.Context(0)
    ctx.set_param("filter_mfma", 1)
    cols = [np.ascontiguousarray(xyz[:, a]) for a in range(3)]
    d = [ctx.alloc(4 * n).upload(c) for c in cols]
    out = ctx.alloc(4 * n)
    info = ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, n, 0, n, k, out.ptr, algo=GRID, want_info=True)
    got = out.download(np.float32, n)
    assert _explain(got, ref) == "ok", info
    for a in d + [out]:
        a.free()
    ctx.close()


def test_outlier_inflated_bounding_box_is_refined(lib):
    """the realistic failure mode of a uniform grid: 0.5 % far floaters make the bounding box 10^6 x the volume of
    the scene, the whole scene falls into one cell, and a single-level search would be quadratic (minutes at 1M).
    The host entry point re-runs the deferred bricks on a finer grid; result identical to cKDTree."""
    xyz = datasets.scene_with_floaters(150_000, 3)
    ref = osor.sor(xyz, 16, 1.0)
    import time
    t0 = time.perf_counter()
    res = lib.sor_filter(xyz, 16, 1.0, want_info=True)
    dt = time.perf_counter() - t0
    assert _explain(res["mean_dists"], ref["mean_dists"]) == "ok", res["info"]
    np.testing.assert_array_equal(res["mask"], ref["mask"])
    assert res["info"]["n_deferred_bricks"] > 0 and res["info"]["n_refined"] > 140_000, res["info"]
    assert dt < 1.0, "refinement did not kick in: %.2f s" % dt  # single level: ~3 s of GPU time at this size
    # the same cloud through the asynchronous device API with the knob off and on
    ctx = lib.Context(0)
    small = datasets.scene_with_floaters(30_000, 4)
    rs = osor.mean_dists_ckdtree(small, 16)
    cols = [np.ascontiguousarray(small[:, a]) for a in range(3)]
    d = [ctx.alloc(4 * len(small)).upload(c) for c in cols]
    out = ctx.alloc(4 * len(small))
    for ad in (0, 1):
        ctx.set_param("adaptive", ad)
        info = ctx.sor_knn(d[0].ptr, d[1].ptr, d[2].ptr, 1, len(small), 0, len(small), 16, out.ptr, algo=GRID, want_info=True)
        assert _explain(out.download(np.float32, len(small)), rs) == "ok", (ad, info)
        assert (info["n_deferred_bricks"] > 0) == bool(ad)
    for a in d + [out]:
        a.free()
    ctx.close()


def test_stats_kernel_matches_numpy(lib):
    """gsx_sor_stats_dev == np.mean / np.std / threshold, bit for bit, at ragged sizes"""
    rng = np.random.default_rng(9)
    ctx = lib.Context(0)
    for n in (1, 5, 8, 127, 128, 129, 1000, 8191, 8192, 8193, 16385, 70001, 1000003):
        a = (rng.random(n, dtype=np.float32) * 0.3 + 0.01).astype(np.float32)
        if n > 100:
            a[rng.integers(0, n, 5)] *= 40
        d = ctx.alloc(4 * n).upload(a)
        st = ctx.alloc(16)
        mk = ctx.alloc(n + 4)
        for tf in (1.0, 10.5, 12.444444444444445):
            ctx.sor_stats(d.ptr, n, tf, st.ptr)
            ctx.sor_mask(d.ptr, n, st.ptr + 8, mk.ptr)
            got = st.download(np.float32, 3)
            want = osor.threshold_numpy(a, tf)
            assert all(np.float32(g).tobytes() == np.float32(w).tobytes() for g, w in zip(got, want)), (n, tf, got, want)
            mask = mk.download(np.uint8, n).view(np.bool_)
            np.testing.assert_array_equal(mask, a < want[2])
        for x in (d, st, mk):
            x.free()
    ctx.close()


def test_sor_10m_subset_and_properties(lib):
    """BASELINE.json configs[2] size: 10M splats k=16.  Full-size checks that do not need a full CPU run:
    a random 4000-query subset against the scalar C restatement, plus sanity properties."""
    n = 10_000_000
    xyz = datasets.uniform(n, 5.0, 0)
    res = lib.sor_filter(xyz, 16, 1.0, algo=GRID, want_info=True)
    md = res["mean_dists"]
    assert np.isfinite(md).all() and (md > 0).all()
    rng = np.random.default_rng(1)
    q = np.sort(rng.choice(n, 600, replace=False))
    ref = osor.mean_dists_brute_subset_c(xyz, 16, q)
    assert _explain(md[q], ref) == "ok", res["info"]
    m, s, t = osor.threshold_numpy(md, 1.0)
    assert np.float32(t).tobytes() == np.float32(res["threshold"]).tobytes()
    np.testing.assert_array_equal(res["mask"], md < t)
    assert res["info"]["n_fallback"] < 0.05 * n


def test_drop_in_dataprocessor(gsx, golden_cases, golden_arrays):
    """the reference-shaped API: DataProcessor(data).remove_flyers(k, sigma) returns the filtered array"""
    case = golden_cases["sor"]["sor_u100k_k8_s1"]
    xyz = datasets.make(case["dataset"])
    arr = np.zeros(len(xyz), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("opacity", "f4")])
    arr["x"], arr["y"], arr["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    arr["opacity"] = np.arange(len(xyz))
    proc = gsx.DataProcessor(arr)
    out = proc.remove_flyers(8, 1.0)
    assert out is proc.data and len(out) == case["survivors"] == 84668
    mask = np.unpackbits(golden_arrays["sor_u100k_k8_s1__mask"])[:len(xyz)].astype(bool)
    np.testing.assert_array_equal(out["opacity"], arr["opacity"][mask])
    with pytest.raises(TypeError):
        gsx.DataProcessor([1, 2, 3]).remove_flyers()
    mask2 = gsx.gpu_ops.filter_sor_gpu(xyz, 8, 1.0)
    np.testing.assert_array_equal(mask2, mask)
    with pytest.raises(ValueError):
        gsx.gpu_ops.filter_sor_gpu(np.zeros((10, 2), np.float32))


def test_non_finite_coordinates_fail_loudly(gsx, lib):
    xyz = datasets.uniform(5000, 10.0, 3)
    xyz[17, 1] = np.nan
    with pytest.raises(gsx.GsxError, match="finite"):
        lib.sor_filter(xyz, 16, 1.0, algo=GRID)
    xyz[17, 1] = np.inf
    with pytest.raises(gsx.GsxError, match="finite"):
        lib.sor_filter(xyz, 16, 1.0, algo=GRID)


def test_clustered_cloud_with_heavy_cells(lib):
    """very uneven density: most points sit in a handful of grid cells (bricks with thousands of
    query batches go through the second knn_brick launch)"""
    xyz = datasets.clustered(120000, seed=11)
    ref = osor.mean_dists_ckdtree(xyz, 16)
    res = lib.sor_filter(xyz, 16, 2.0, algo=GRID, want_info=True)
    assert _explain(res["mean_dists"], ref) == "ok", res["info"]


def test_degenerate_geometries(lib):
    rng = np.random.default_rng(5)
    n = 20000
    flat = np.stack([rng.random(n) * 10, rng.random(n) * 10, np.full(n, 3.0)], 1).astype(np.float32)   # plane
    line = np.stack([rng.random(n) * 10, np.full(n, 1.0), np.full(n, -2.0)], 1).astype(np.float32)     # line
    same = np.tile(np.array([[1.5, 2.5, 3.5]], np.float32), (3000, 1))                                  # one point
    for name, xyz in (("plane", flat), ("line", line), ("identical", same)):
        ref = osor.mean_dists_ckdtree(xyz, 16)
        for algo in (BRUTE, GRID):
            res = lib.sor_filter(xyz, 16, 1.0, algo=algo, want_info=True)
            assert _explain(res["mean_dists"], ref) == "ok", (name, algo, res["info"])


_TORCH_SCRIPT = r"""
import importlib, json, sys
import numpy as np
import torch                      # FIRST: libgsx_hip.so must bind to the HIP runtime torch bundles
sys.path.insert(0, sys.argv[1])
gdist = importlib.import_module("3dgsconverter_amd.dist")
from oracle import datasets
spec = json.loads(sys.argv[2])
xyz = datasets.make(spec["dataset"])
t = torch.from_numpy(xyz).to(torch.device("cuda", 0))
res = gdist.sharded_sor(t, spec["k"], spec["sigma"], gdist.HipCompute(0))
torch.cuda.synchronize()
np.save(sys.argv[3], res.mask_local.cpu().numpy())
np.save(sys.argv[4], np.concatenate([res.stats.cpu().numpy(), res.mean_dists_local.cpu().numpy()]))
"""


def test_torch_plumbing_and_sharded_entry_point(golden_cases, golden_arrays, tmp_path):
    """bench.py's path: torch tensors in HBM -> dist.sharded_sor(HipCompute) -> mask, single process
    (world 1).  Runs in its own interpreter because torch must be imported BEFORE libgsx_hip.so (both
    link a libamdhip64.so.7; whichever loads first serves both, and torch only works with its own)."""
    import json
    import os
    import subprocess
    import sys
    pytest.importorskip("torch")
    case = golden_cases["sor"]["sor_u100k_k8_s1"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = json.dumps({"dataset": case["dataset"], "k": case["k_used"], "sigma": case["sigma_used"]})
    m, o = str(tmp_path / "mask.npy"), str(tmp_path / "out.npy")
    r = subprocess.run([sys.executable, "-c", _TORCH_SCRIPT, root, spec, m, o], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    mask = np.load(m).astype(bool)
    out = np.load(o)
    np.testing.assert_array_equal(np.packbits(mask), golden_arrays["sor_u100k_k8_s1__mask"])
    assert np.float32(out[2]).tobytes() == f32_from_hex(case["threshold_hex"]).tobytes()
    assert sha16(out[3:].astype(np.float32).tobytes()) == case["mean_dists_sha"]


def test_sor_50m_k32_config4_single_gpu(lib):
    """BASELINE.json configs[3] size on ONE GPU (it fits 288 GB easily): 50M splats, k=32.  Checks a
    random query subset against the scalar C restatement and the mask against numpy on the GPU's own
    mean distances (the statistics are bit-exact, so the mask must be identical)."""
    n = 50_000_000
    xyz = datasets.uniform(n, 10.0, 0)
    res = lib.sor_filter(xyz, 32, 1.0, algo=GRID, want_info=True)
    md = res["mean_dists"]
    assert np.isfinite(md).all() and (md > 0).all()
    q = np.sort(np.random.default_rng(2).choice(n, 48, replace=False))
    ref = osor.mean_dists_brute_subset_c(xyz, 32, q)
    assert _explain(md[q], ref) == "ok", res["info"]
    m, s, t = osor.threshold_numpy(md, 1.0)
    assert np.float32(t).tobytes() == np.float32(res["threshold"]).tobytes()
    np.testing.assert_array_equal(res["mask"], md < t)
    assert res["info"]["n_fallback"] < 0.02 * n
