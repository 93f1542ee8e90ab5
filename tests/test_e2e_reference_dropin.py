"""not gpu, build container only: the REFERENCE's own orchestrator and SOG writer driven through install().

``install()`` rebinds the names ``converter.py:10`` and ``sog.py:11`` resolve; these tests run the reference's
``Converter.run`` (parquet -> parquet, no plyfile needed) with ``--density_sensitivity`` / ``--sor_k --sor_sigma`` /
``--sor_intensity`` and its ``SogFormat.write`` with ``--compression_level 2`` and check that

  * every hot-path call reaches the product's entry points (the C-ABI wrappers in ``3dgsconverter_amd._lib``),
  * the output equals what the UN-patched reference produces for the same input (density: file contents
    identical; SOR: the reference's CPU branch computes its mask and forgets to apply it, SURVEY.md F3, so the
    expectation is its own captured mask applied to its own output; SOG: every texture that does not depend on
    the random K-Means init is identical, the clustered ones are consistent with the written codebooks).

There is no GPU here, so the five GPU entry points of ``_lib`` are replaced by the ORACLE (the checker standing in
for the device, allowed in tests/ only) and record that they were hit; the GPU kernels themselves are compared
with the same oracle / fixtures in the ``-m gpu`` tests.  The host-side C routines (row gather / compaction) are
the real ones.  /root/reference does not exist on the GPU box: skipped there.
"""
import importlib
import io
import json
import os
import zipfile

import numpy as np
import pytest

from oracle import datasets, density as oden, kmeans as okm, refload, sor as osor

pytestmark = pytest.mark.skipif(not refload.available(), reason="reference not mounted (build container only)")


@pytest.fixture()
def dropin(gsx, monkeypatch):
    """install() with the device entry points served by the oracle; yields the call log."""
    refload.load()
    lib = gsx._lib
    hits = []

    def sor_filter(xyz, k, threshold_factor, algo=0, want_mean=True, want_info=False):
        hits.append(("gsx_sor_filter", len(xyz), int(k), float(threshold_factor)))
        r = osor.sor(np.ascontiguousarray(xyz, dtype=np.float32), int(k), float(threshold_factor))
        return {"mask": r["mask"], "mean_dists": r["mean_dists"] if want_mean else None, "mean": r["mean"],
                "std": r["std"], "threshold": r["threshold"], "info": None}

    def density_voxels(xyz, voxel_size, min_points, dense_cap=None):
        hits.append(("gsx_density_voxels", len(xyz), float(voxel_size), int(min_points)))
        keys = oden.voxel_keys(np.asarray(xyz), voxel_size)
        uniq, counts = np.unique(keys, axis=0, return_counts=True)
        dense = counts >= min_points
        return {"n_unique": len(uniq), "dense_keys": uniq[dense], "dense_counts": counts[dense]}

    def density_mask(xyz, voxel_size, kept_keys):
        hits.append(("gsx_density_mask", len(xyz), len(kept_keys)))
        kept = set(map(tuple, np.asarray(kept_keys).reshape(-1, 3).tolist()))
        keys = oden.voxel_keys(np.asarray(xyz), voxel_size)
        return np.fromiter((tuple(k) in kept for k in keys.tolist()), dtype=bool, count=len(keys))

    def kmeans_lloyd(data, init, max_iter):
        hits.append(("gsx_kmeans_lloyd", data.shape, init.shape[0], int(max_iter)))
        c, l, _ = okm.lloyd(data, init, int(max_iter))
        return c, l

    def kmeans_lloyd_many(problems, max_iter, lanes=8, device=0):
        # the palette's chunks on concurrent contexts: same per-chunk entry point (gsx_kmeans_lloyd_dev), same results
        return [kmeans_lloyd(d, i, max_iter) for d, i in problems]

    def quantize(vals, cb):
        hits.append(("gsx_quantize_sorted_codebook", len(vals), len(cb)))
        return okm.quantize_to_codebook(np.asarray(vals, np.float32), np.asarray(cb, np.float32))

    class FakeChain:
        """_lib.DeviceChain (coordinates resident in HBM across filters) with the oracle as the device"""

        def __init__(self, xyz_rows=None, device=0, table=None):
            if table is not None:      # round 6: the class gathers the coordinates itself (into a page-locked staging buffer)
                xyz_rows = lib.host_gather_xyz(table)
            self.xyz = np.ascontiguousarray(xyz_rows, dtype=np.float32)
            self.n0 = self.n = len(self.xyz)
            self.idx = np.arange(self.n0)
            hits.append(("DeviceChain", self.n0))

        def density_voxels(self, voxel_size, min_points):
            hits.append(("gsx_density_voxels_dev", self.n, float(voxel_size), int(min_points)))
            return density_voxels(self.xyz, voxel_size, min_points)

        def density_filter(self, voxel_size, min_points, keep_multicluster):
            """gsx_density_filter_dev: the whole filter in one device call (the oracle, pinned to the reference, stands in)"""
            from oracle import density as oden
            hits.append(("gsx_density_filter_dev", self.n, float(voxel_size), int(min_points)))
            keys = oden.voxel_keys(self.xyz, voxel_size)
            uniq, inv, counts = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
            dense = uniq[counts >= min_points]
            out = {"status": 0, "n_unique": len(uniq), "kept_clusters": 0, "largest": 0, "left": None}
            if len(dense) == 0:
                out["status"] = 1
                return out
            import importlib
            cl = importlib.import_module("3dgsconverter_amd.processing.clusters")
            comps = cl.connected_clusters(map(tuple, dense.tolist()))
            kept, out["kept_clusters"], out["largest"] = cl.select_clusters(comps, keep_multicluster)
            mask = np.fromiter((tuple(k) in kept for k in keys.tolist()), dtype=bool, count=len(keys))
            out["left"] = self._keep(mask)
            return out

        def _keep(self, mask):
            self.xyz, self.idx = self.xyz[mask], self.idx[mask]
            self.n = len(self.idx)
            return self.n

        def density_keep(self, voxel_size, kept_keys):
            hits.append(("gsx_density_mask_dev", self.n, len(kept_keys)))
            return self._keep(density_mask(self.xyz, voxel_size, kept_keys))

        def keep_none(self):
            self._keep(np.zeros(self.n, bool))

        def bbox_keep(self, b):
            hits.append(("gsx_mask_bbox_dev", self.n))
            b = [np.float32(v) if type(v) in (float, int) else v for v in b]
            x, y, z = self.xyz.T
            return self._keep((x >= b[0]) & (x <= b[3]) & (y >= b[1]) & (y <= b[4]) & (z >= b[2]) & (z <= b[5]))

        def ge_keep(self, column, threshold):
            hits.append(("gsx_mask_ge_dev", self.n))
            return self._keep(column[self.idx].astype(np.float64) >= threshold)

        def sor_keep(self, k, threshold_factor):
            hits.append(("gsx_sor_knn_dev", self.n, int(k), float(threshold_factor)))
            r = osor.sor(self.xyz, int(k), float(threshold_factor))
            return {"mean": r["mean"], "std": r["std"], "threshold": r["threshold"], "kept": self._keep(r["mask"])}

        def survivors(self):
            return self.idx.astype(np.uint32)

        def bbox(self):
            hits.append(("gsx_slab_bbox_dev", self.n))
            return self.xyz.min(0).tolist(), self.xyz.max(0).tolist()

        def close(self):
            pass

    def lexsort3(k0, k1, k2):
        hits.append(("gsx_lexsort3", len(k0)))
        return np.lexsort((k0, k1, k2))

    def sog_quats(rows):
        from oracle import sog as osog
        hits.append(("gsx_sog_quats", len(rows)))
        ds = np.zeros(len(rows), dtype=[("rot_%d" % c, "f4") for c in range(4)])
        for c in range(4):
            ds["rot_%d" % c] = rows[:, c]
        return osog.quats(ds)

    def sog_positions(v, stats=None):
        hits.append(("gsx_sog_positions", len(v)))
        t = np.sign(v) * np.log(np.abs(v) + 1.0)                      # formats/sog.py:280-295 on one axis
        mn, mx = np.min(t), np.max(t)
        return np.clip((t - mn) / (mx - mn) * 65535, 0, 65535).astype(np.uint16), mn, mx

    def sog_alpha(o, stats=None):
        hits.append(("gsx_sog_alpha", len(o)))
        return np.clip(1.0 / (1.0 + np.exp(-o)) * 255, 0, 255).astype(np.uint8)   # formats/sog.py:457-459

    def rgb_from_sh(f_dc, stats=None):
        hits.append(("gsx_rgb_from_sh", len(f_dc)))
        lin = np.clip(0.5 + np.asarray(f_dc, np.float32) * 0.28209479177387814, 0.0, 1.0)   # data_processor.py:332-342
        return (np.power(lin, 1.0 / 2.2) * 255).astype(np.uint8)

    def kmeans1d(vals, k, iters=50, want_labels=False, want_inertia=False):
        hits.append(("gsx_kmeans1d", len(vals), int(k), int(iters)))
        cent, inertia = okm.kmeans1d_sorted(vals, int(k), int(iters))
        out = (cent,)
        if want_labels:
            out += (okm.assign(np.asarray(vals, np.float32).reshape(-1, 1), cent.reshape(-1, 1)),)
        if want_inertia:
            out += (inertia,)
        return out[0] if len(out) == 1 else out

    def morton_order(x, y, z, ctx=None, keep_device=False):
        from oracle import cply as ocply
        hits.append(("gsx_morton_order_dev", len(x)))
        return ocply.morton_order(x, y, z)

    def cply_pack(columns, order, sh_columns=(), ctx=None):
        from oracle import cply as ocply
        hits.append(("gsx_cply_pack_dev", len(order), len(sh_columns)))
        n = len(order)
        d = np.zeros(n, dtype=[(c, "f4") for c in lib.CPLY_COLUMNS if c != "alpha"] + [("opacity", "f4")] +
                     [("f_rest_%d" % i, "f4") for i in range(len(sh_columns))])
        for c in lib.CPLY_COLUMNS:
            if c != "alpha":
                d[c] = columns[c]
        a = np.asarray(columns["alpha"], dtype=np.float32)
        with np.errstate(divide="ignore"):
            d["opacity"] = -np.log(1.0 / a - 1.0)       # only used to rebuild alpha below; replaced right after
        for i, col in enumerate(sh_columns):
            d["f_rest_%d" % i] = col
        chunks, verts, sh = ocply.encode(d, order, ["f_rest_%d" % i for i in range(len(sh_columns))])
        na = np.clip(np.floor(a[order] * 255 + 0.5), 0, 255).astype(np.uint32)   # the alpha byte from the alpha actually given
        verts[:, 3] = (verts[:, 3] & 0xffffff00) | na
        return chunks, verts, sh

    def cply_pack_table(data, sh_names, order=None, ctx=None):
        # round 5: the writer hands the whole table over (one gather, one upload); the stand-in is the oracle's own encode
        from oracle import cply as ocply
        levels = None
        if order is None:
            hits.append(("gsx_morton_order_dev", len(data)))
            order, levels = ocply.morton_order(data["x"], data["y"], data["z"])
        if callable(sh_names):      # round 6: the writer's degree detection arrives as a function (None = scan the columns on the host)
            sh_names = sh_names(None)
        hits.append(("gsx_cply_pack_dev", len(order), len(sh_names)))
        chunks, verts, sh = ocply.encode(data, order, list(sh_names))
        return chunks, verts, sh, order, levels

    class FakeCtx:
        def __init__(self, device=0):
            pass

        def close(self):
            pass

    for name, fn in (("morton_order", morton_order), ("cply_pack", cply_pack), ("cply_pack_table", cply_pack_table), ("Context", FakeCtx), ("sor_filter", sor_filter), ("density_voxels", density_voxels), ("density_mask", density_mask),
                     ("kmeans_lloyd", kmeans_lloyd), ("kmeans_lloyd_many", kmeans_lloyd_many), ("quantize_sorted_codebook", quantize), ("DeviceChain", FakeChain),
                     ("lexsort3", lexsort3), ("sog_quats", sog_quats),
                     ("sog_positions", sog_positions), ("sog_alpha", sog_alpha), ("kmeans1d", kmeans1d),
                     ("rgb_from_sh", rgb_from_sh), ("require_hip", lambda: None)):
        monkeypatch.setattr(lib, name, fn)
    monkeypatch.setattr(gsx.gpu_ops, "HAS_HIP", True)
    monkeypatch.setattr(gsx.gpu_ops, "HAS_TAICHI", True)
    # the SOG writer's device-resident core (formats/sog_device.py) is one sequence of _dev calls on a resident table: it has no
    # per-stage host entry points for the oracle to stand in for, so this GPU-less test drives the host-staged core (the same
    # kernels' arithmetic, one call per stage); the resident core is tested on the GPU against the reference's own bundle
    # (tests/test_sog_gpu.py)
    import importlib
    monkeypatch.setattr(importlib.import_module("3dgsconverter_amd.formats.sog_writer"), "DEVICE_RESIDENT", False)
    gsx.install()
    try:
        yield hits
    finally:
        gsx.uninstall()


def _write_input(tmp_path, n=6000, seed=3):
    """a parquet file in the reference's own column naming, written by the reference's own codec"""
    refload.load()
    from gsconverter.formats.parquet import ParquetFormat
    data = datasets.sog_scene(n, seed)
    # a compact core + a sparse halo, so that both filters remove something
    rng = np.random.default_rng(seed)
    far = rng.random(n) < 0.06
    for ax in "xyz":
        data[ax] = np.where(far, data[ax] * np.float32(9.0), data[ax] * np.float32(0.4))
    path = str(tmp_path / "in.parquet")
    ParquetFormat().write(data, path)
    return path, data


def _run(tmp_path, inp, tag, **kw):
    from gsconverter.converter import Converter
    import pandas as pd
    out = str(tmp_path / ("out_%s.parquet" % tag))
    Converter(inp, out, "parquet").run(**kw)
    return pd.read_parquet(out)


def test_converter_run_density_flag_goes_through_the_dropin(tmp_path, gsx, dropin):
    inp, _ = _write_input(tmp_path)
    import gsconverter.converter as conv
    assert issubclass(conv.DataProcessor, gsx.DataProcessor)
    got = _run(tmp_path, inp, "dropin", density_sensitivity=0.3)
    # install() binds the chained class: one upload, the filters' device entry points, one compaction at `.data`
    assert [h[0] for h in dropin] == ["DeviceChain", "gsx_density_filter_dev"]   # the whole filter: ONE device call
    gsx.uninstall()
    assert not issubclass(conv.DataProcessor, gsx.DataProcessor)
    want = _run(tmp_path, inp, "reference", density_sensitivity=0.3)
    assert 0 < len(want) < 6000
    assert got.equals(want)


def test_install_binds_the_lazy_class_to_the_orchestrator_only(gsx, dropin):
    """callers of the PUBLIC class use the filters' return values (the reference returns self.data),
    so gsconverter.processing.DataProcessor stays the eager class; only converter.py's name -- whose return values are
    ignored (converter.py:196-236) -- gets the lazy chain."""
    import gsconverter.converter as conv
    import gsconverter.processing as rp
    import gsconverter.processing.data_processor as rdp
    dp = importlib.import_module("3dgsconverter_amd.processing.data_processor")
    assert conv.DataProcessor is dp.ChainedDataProcessor
    assert rp.DataProcessor is dp.DataProcessor and rdp.DataProcessor is dp.DataProcessor
    xyz = datasets.uniform(4000, 3.0, 1)
    arr = np.zeros(len(xyz), dtype=[("x", "f4"), ("y", "f4"), ("z", "f4")])
    arr["x"], arr["y"], arr["z"] = xyz.T
    proc = rp.DataProcessor(arr)
    out = proc.apply_density_filter(voxel_size=1.0, threshold_percentage=0.5)
    assert out is proc.data and len(out) > 0          # eager: returns the table, like data_processor.py:117
    out2 = proc.remove_flyers(k=8, threshold_factor=1.0)
    assert out2 is proc.data and 0 < len(out2) <= len(out)


def test_converter_run_sor_flags_go_through_the_dropin(tmp_path, gsx, dropin):
    inp, _ = _write_input(tmp_path)
    got = _run(tmp_path, inp, "dropin", sor_k=12.0, sor_sigma=1.5)  # main.py parses --sor_k as float
    assert dropin == [("DeviceChain", 6000), ("gsx_sor_knn_dev", 6000, 12, 1.5)]
    got_i = _run(tmp_path, inp, "dropin_i", sor_intensity=7)
    assert dropin[3][:2] == ("gsx_sor_knn_dev", 6000) and dropin[3][2] == int(10 + 6 * (40 / 9))
    gsx.uninstall()
    # the un-patched reference: mask computed (data_processor.py:180) but not applied (:181-182)
    ref_all = _run(tmp_path, inp, "reference", sor_k=12.0, sor_sigma=1.5)
    assert len(ref_all) == 6000
    xyz = np.column_stack([ref_all["x"], ref_all["y"], ref_all["z"]]).astype(np.float32)
    cap = refload.reference_sor(xyz, 12, 1.5)
    assert 0 < cap["mask"].sum() < 6000
    assert got.reset_index(drop=True).equals(ref_all[cap["mask"]].reset_index(drop=True))
    cap_i = refload.reference_sor(xyz, 25, 10.5, intensity=7)
    assert got_i.reset_index(drop=True).equals(ref_all[cap_i["mask"]].reset_index(drop=True))


def test_converter_run_density_then_sor_stays_on_the_device(tmp_path, gsx, dropin):
    """BASELINE.json configs[2] through the reference's CLI path: --density_sensitivity + --sor_k/--sor_sigma.  ONE chain:
    the second filter runs on the rows the first one left in HBM, the host table is compacted once."""
    inp, _ = _write_input(tmp_path)
    got = _run(tmp_path, inp, "dropin", density_sensitivity=0.3, sor_k=10.0, sor_sigma=1.0)
    names = [h[0] for h in dropin]
    assert names.count("DeviceChain") == 1 and names[-1] == "gsx_sor_knn_dev"
    n_after_density = dropin[-1][1]
    gsx.uninstall()
    dens = _run(tmp_path, inp, "reference_density", density_sensitivity=0.3)
    assert len(dens) == n_after_density
    xyz = np.column_stack([dens["x"], dens["y"], dens["z"]]).astype(np.float32)
    cap = refload.reference_sor(xyz, 10, 1.0)   # the reference's own SOR mask on its own density output
    assert got.reset_index(drop=True).equals(dens[cap["mask"]].reset_index(drop=True))


def test_converter_run_all_four_filters_are_one_chain(tmp_path, gsx, dropin):
    """--bbox, --min_opacity, --density_sensitivity, --sor_k/--sor_sigma in the orchestrator's order (converter.py:196-236):
    four masks over the device-resident rows, one chain, one host compaction; same table as the reference's own run with
    its (computed, not applied) SOR mask applied"""
    inp, _ = _write_input(tmp_path)
    box = (-2.5, -3.0, -2.0, 3.0, 2.75, 2.2)
    got = _run(tmp_path, inp, "dropin", bbox=box, min_opacity=40, density_sensitivity=0.3, sor_k=10.0, sor_sigma=1.0)
    names = [h[0] for h in dropin]
    assert names == ["DeviceChain", "gsx_mask_bbox_dev", "gsx_mask_ge_dev", "gsx_density_filter_dev", "gsx_sor_knn_dev"]
    gsx.uninstall()
    pre = _run(tmp_path, inp, "reference_pre", bbox=box, min_opacity=40, density_sensitivity=0.3)
    assert 0 < len(pre) < 6000 and len(pre) == dropin[-1][1]
    xyz = np.column_stack([pre["x"], pre["y"], pre["z"]]).astype(np.float32)
    cap = refload.reference_sor(xyz, 10, 1.0)
    assert got.reset_index(drop=True).equals(pre[cap["mask"]].reset_index(drop=True))


def test_converter_run_sh_cap_and_rgb_are_the_dropins_own(tmp_path, gsx, dropin):
    """--sh_level 1 --rgb through the orchestrator (converter.py:188,252): cap_sh_degree and add_rgb_from_sh are methods of the
    drop-in, not forwarded to the reference class; same table as the un-patched reference writes --
    also combined with filters, where the lazy class defers the column fill until the table has been compacted"""
    inp, _ = _write_input(tmp_path)
    dp = importlib.import_module("3dgsconverter_amd.processing.data_processor")
    for attr in ("cap_sh_degree", "add_rgb_from_sh", "apply_auto_bbox", "_compute_rgb_from_sh"):
        assert attr in vars(dp.DataProcessor), attr
    got = _run(tmp_path, inp, "dropin", sh_level=1, rgb=True, auto_bbox=True)
    assert [h for h in dropin] == [("gsx_rgb_from_sh", 3 * 6000)]        # round 6: the three channels in one call
    del dropin[:]
    got_f = _run(tmp_path, inp, "dropin_f", sh_level=0, rgb=True, density_sensitivity=0.3, auto_bbox=True)
    assert [h[0] for h in dropin if h[0].endswith("_dev") or "rgb" in h[0]] == \
        ["gsx_density_filter_dev", "gsx_slab_bbox_dev"] + ["gsx_rgb_from_sh"]
    gsx.uninstall()
    want = _run(tmp_path, inp, "reference", sh_level=1, rgb=True, auto_bbox=True)
    zero_cols = int((want.filter(regex="_sh").abs().sum() == 0).sum())     # the parquet codec's r/g/b_sh* columns
    assert {"red", "green", "blue"} <= set(want.columns) and zero_cols == 36   # f_rest_9..44 zeroed, 3 DC + 9 AC kept
    assert got.equals(want)
    want_f = _run(tmp_path, inp, "reference_f", sh_level=0, rgb=True, density_sensitivity=0.3, auto_bbox=True)
    assert 0 < len(want_f) < 6000 and int((want_f.filter(regex="_sh").abs().sum() == 0).sum()) == 45
    assert got_f.equals(want_f)


def test_compressed_ply_writer_goes_through_the_dropin(tmp_path, gsx, dropin):
    """install() rebinds CompressedPlyFormat.write: Morton order, chunk bounds and packers reach the product's entry points;
    the file holds exactly the elements the reference's own writer produces (with its argsort made stable)"""
    refload.load()
    from gsconverter.converter import Converter
    data = ocply_scene()
    inp = str(tmp_path / "in.parquet")
    from gsconverter.formats.parquet import ParquetFormat
    ParquetFormat().write(data, inp)
    out = str(tmp_path / "out.compressed.ply")
    Converter(inp, out, "compressed_ply").run()
    assert [h[0] for h in dropin] == ["gsx_morton_order_dev", "gsx_cply_pack_dev"] and dropin[1][1:] == (len(data), 45)
    gsx.uninstall()
    table = ParquetFormat().read(inp)
    ref = refload.reference_cply(table, stable_ties=True)
    raw = open(out, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    nc = len(ref["chunk"])
    assert ("element chunk %d" % nc).encode() in head and ("element vertex %d" % len(data)).encode() in head
    assert body == ref["chunk"].tobytes() + ref["vertex"].tobytes() + ref["sh"].tobytes()


def ocply_scene():
    from oracle import cply as ocply
    return ocply.cply_scene(2500, 12, "clustered")


def _decode(path):
    from PIL import Image
    with zipfile.ZipFile(path) as zf:
        meta = json.loads(zf.read("meta.json"))
        tex = {n[:-5]: np.asarray(Image.open(io.BytesIO(zf.read(n))).convert("RGBA"), dtype=np.uint8).reshape(-1, 4)
               for n in zf.namelist() if n.endswith(".webp")}
    return meta, tex


def test_sog_writer_goes_through_the_dropin(tmp_path, gsx, dropin):
    """install() rebinds SogFormat.write to formats/sog_writer.py:write_sog: spatial sort, quaternion packing, codebooks,
    quantiser and SH palette reach the product's entry points; every texture that does not depend on the random K-Means
    init is byte-identical to what the un-patched reference writes from the same table"""
    refload.load()
    import gsconverter.formats.sog as sogmod
    assert sogmod.gpu_ops is gsx.gpu_ops
    n = 4000
    data = datasets.sog_scene(n, 8)
    a = str(tmp_path / "dropin.sog")
    np.random.seed(1)
    sogmod.SogFormat().write(data, a, compression_level=2)
    names = [h[0] for h in dropin]
    assert names[0] == "gsx_lexsort3" and names[1:4] == ["gsx_sog_positions"] * 3 and names[4] == "gsx_sog_quats"
    assert names.count("gsx_sog_alpha") == 1
    plan = okm.sog_sh_plan(n, 2)
    km = [h for h in dropin if h[0] == "gsx_kmeans_lloyd"]
    k1 = [h for h in dropin if h[0] == "gsx_kmeans1d"]
    sizes = [min(plan["chunk_size"], n - i * plan["chunk_size"]) for i in range(plan["num_chunks"])]
    # two scalar codebooks (sog.py:402,443) and the palette's codebook (:561) through the scalar solver ...
    assert [h[1:3] for h in k1] == [(3 * n, 256), (3 * n, 256), (45 * sum(min(s, plan["k_per_chunk"]) for s in sizes), 256)]
    # ... and one Lloyd call per SH chunk (:544); the K >= N shortcut never reaches the device
    want_chunks = [((s, 45), min(s, plan["k_per_chunk"]), 10) for s in sizes if min(s, plan["k_per_chunk"]) < s]
    assert [h[1:] for h in km] == want_chunks
    # the quantiser is reachable now: 3 scale columns + 3 colour columns + the palette's centroid scalars
    qz = [h for h in dropin if h[0] == "gsx_quantize_sorted_codebook"]
    assert [h[1] for h in qz[:6]] == [n] * 6 and len(qz) == 7 and qz[6][1] == 45 * sum(min(s, plan["k_per_chunk"]) for s in sizes)
    gsx.uninstall()
    b = str(tmp_path / "reference.sog")
    np.random.seed(1)
    sogmod.SogFormat().write(data, b, compression_level=2)
    ma, ta = _decode(a)
    mb, tb = _decode(b)
    for name in ("means_l", "means_u", "quats"):
        np.testing.assert_array_equal(ta[name], tb[name])
    np.testing.assert_array_equal(ta["sh0"][:, 3], tb["sh0"][:, 3])  # opacity byte
    assert ma["means"] == mb["means"] and ma["count"] == mb["count"] == n
    assert ma["shN"]["count"] == mb["shN"]["count"] and ma["shN"]["bands"] == 3
    assert {k: v["files"] for k, v in ma.items() if isinstance(v, dict) and "files" in v} == \
           {k: v["files"] for k, v in mb.items() if isinstance(v, dict) and "files" in v}
    assert ta["shN_centroids"].shape == tb["shN_centroids"].shape and ta["shN_labels"].shape == tb["shN_labels"].shape
    # the clustered textures are consistent with the codebooks written next to them (sog.py:408-423,447-449)
    ds = data[np.lexsort((data["z"], data["y"], data["x"]))]
    for tex, cols in (("scales", ["scale_0", "scale_1", "scale_2"]), ("sh0", ["f_dc_0", "f_dc_1", "f_dc_2"])):
        cb = np.array(ma[tex]["codebook"], dtype=np.float32)
        vis = ta[tex][:n, 3] != 0
        for ch, col in enumerate(cols):
            np.testing.assert_array_equal(okm.quantize_to_codebook(ds[col], cb)[vis], ta[tex][:n, ch][vis])
    # every splat's palette label points at a valid centroid; labels of a chunk stay inside the chunk's slice of the palette
    lab = ta["shN_labels"][:n, 0].astype(np.int64) + 256 * ta["shN_labels"][:n, 1].astype(np.int64)
    assert lab.max() < ma["shN"]["count"]
    # codebook QUALITY: the un-patched run used the reference's sklearn path (k-means++-seeded MiniBatchKMeans); the drop-in's
    # scalar solver must not be worse on any of the three codebooks
    for tex, cols in (("scales", ["scale_0", "scale_1", "scale_2"]), ("sh0", ["f_dc_0", "f_dc_1", "f_dc_2"])):
        flat = np.concatenate([ds[c] for c in cols])
        assert okm.inertia_1d(flat, ma[tex]["codebook"]) <= 1.02 * okm.inertia_1d(flat, mb[tex]["codebook"]), tex
