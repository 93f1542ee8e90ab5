"""-m "not gpu": the multi-GPU choreography (all-gather of xyz, one share of the queries per rank written
into a zero-filled full-length array, sum all-reduce, redundant exact statistics, mask of the local
index range) on CPU with the gloo backend, world_size 2 and 3.
The compute callables are the oracle here (tests only); on the GPU they are the C ABI."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_local, k, sigma, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gdist = importlib.import_module("3dgsconverter_amd.dist")
    from oracle import datasets
    from oracle.slab_backend import GlooHostComm, NumpySlabBackend
    full = datasets.uniform(world * n_local, 10.0, 42)
    be = NumpySlabBackend()
    res = gdist.replicated_sor(be, GlooHostComm(be), be.rows_buffer(full[rank * n_local:(rank + 1) * n_local]), n_local, k, sigma,
                               want_host=True)
    np.save(os.path.join(out_dir, "mask_%d.npy" % rank), res["mask_host"])
    np.save(os.path.join(out_dir, "stats_%d.npy" % rank), res["stats_host"])
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sor_equals_single_process(world, tmp_path):
    """the replicated exchange (dist.replicated_sor: north_star's all-gather design, torch-free in the product)"""
    import torch.multiprocessing as mp
    from oracle import datasets, sor as osor
    n_local, k, sigma = 4000, 16, 1.0
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_local, k, sigma, str(tmp_path)), nprocs=world, join=True)
    full = datasets.uniform(world * n_local, 10.0, 42)
    ref = osor.sor(full, k, sigma, workers=2)
    masks = np.concatenate([np.load(tmp_path / ("mask_%d.npy" % r)) for r in range(world)]).astype(bool)
    np.testing.assert_array_equal(masks, ref["mask"])
    for r in range(world):
        st = np.load(tmp_path / ("stats_%d.npy" % r))
        assert np.float32(st[2]).tobytes() == np.float32(ref["threshold"]).tobytes()


def test_single_process_path_needs_no_process_group():
    import importlib
    from oracle import datasets, sor as osor
    from oracle.slab_backend import NumpySlabBackend
    gdist = importlib.import_module("3dgsconverter_amd.dist")
    xyz = datasets.uniform(3000, 10.0, 1)
    be = NumpySlabBackend()
    res = gdist.replicated_sor(be, None, be.rows_buffer(xyz), len(xyz), 8, 1.0, want_host=True)
    np.testing.assert_array_equal(res["mask_host"], osor.sor(xyz, 8, 1.0, workers=2)["mask"])


def test_multi_gpu_modules_do_not_import_torch():
    """north_star: numpy + ctypes over the C ABI, no PyTorch -- also on the N > 1 path"""
    import subprocess
    code = ("import sys, importlib; sys.path.insert(0, %r);"
            "[importlib.import_module('3dgsconverter_amd.' + m) for m in ('dist', 'dist_slab', 'dist_density', 'dist_palette', 'launch')];"
            "assert 'torch' not in sys.modules, 'torch was imported'" % ROOT)
    subprocess.run([sys.executable, "-c", code], check=True)
    for name in ("dist.py", "dist_slab.py", "dist_density.py", "dist_palette.py", "launch.py"):
        src = open(os.path.join(ROOT, "3dgsconverter_amd", name)).read()
        assert "import torch" not in src, name
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in bench


# ---------------------------------------------------------------------------------------------------------------
# the slab exchange (3dgsconverter_amd/dist_slab.py): partition -> all-to-all -> slab KNN -> certificate ->
# means back -> piece-sum statistics.  gloo + the numpy backend of oracle/slab_backend.py stand in for RCCL + HIP.
def _slab_worker(rank, world, port, n_local, k, sigma, kind, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    from oracle.slab_backend import GlooHostComm, NumpySlabBackend
    full = _slab_cloud(kind, world * n_local)
    be = NumpySlabBackend()
    comm = GlooHostComm(be)
    rows = be.rows_buffer(full[rank * n_local:(rank + 1) * n_local])
    try:
        res = slab.slab_sor(be, comm, rows, n_local, k, sigma, want_host=True)
        np.save(os.path.join(out_dir, "mask_%d.npy" % rank), res["mask_host"])
        np.save(os.path.join(out_dir, "md_%d.npy" % rank), res["mean_dists_host"])
        np.save(os.path.join(out_dir, "stats_%d.npy" % rank), res["stats_host"])
        np.save(os.path.join(out_dir, "sizes_%d.npy" % rank), np.array([res["n_own"], res["n_halo"]]))
    except slab.SlabUncertain as e:
        with open(os.path.join(out_dir, "uncertain_%d.txt" % rank), "w") as f:
            f.write(str(e))
    dist.destroy_process_group()


def _slab_cloud(kind, n):
    from oracle import datasets
    if kind == "uniform":
        return datasets.uniform(n, 10.0, 42)
    if kind == "anisotropic":   # longest axis is y; a density gradient makes equal-count slabs unequal in width
        xyz = datasets.uniform(n, 1.0, 7)
        xyz[:, 1] = (xyz[:, 1] ** 2) * np.float32(40.0)
        xyz[:, 0] *= np.float32(6.0)
        return xyz
    if kind == "floaters":
        return datasets.scene_with_floaters(n, 3)
    raise ValueError(kind)


@pytest.mark.parametrize("world,kind,n_local", [(2, "uniform", 16384), (3, "uniform", 12000), (2, "anisotropic", 12288),
                                                (3, "anisotropic", 9000)])
def test_slab_sor_equals_single_process(world, kind, n_local, tmp_path):
    """mean distances, statistics and masks of the index shards == the un-sharded oracle, bit for bit; the shard sizes
    12000 / 9000 are not multiples of 8192, so numpy's pieces straddle the rank boundaries (head exchange)"""
    import torch.multiprocessing as mp
    from oracle import sor as osor
    k, sigma = 16, 1.0
    mp.spawn(_slab_worker, args=(world, _free_port(), n_local, k, sigma, kind, str(tmp_path)), nprocs=world, join=True)
    full = _slab_cloud(kind, world * n_local)
    ref = osor.sor(full, k, sigma, workers=2)
    assert not list(tmp_path.glob("uncertain_*")), "a uniform-density cloud must be certified inside its slabs"
    md = np.concatenate([np.load(tmp_path / ("md_%d.npy" % r)) for r in range(world)])
    np.testing.assert_array_equal(md.view(np.uint32), ref["mean_dists"].view(np.uint32))
    masks = np.concatenate([np.load(tmp_path / ("mask_%d.npy" % r)) for r in range(world)]).astype(bool)
    np.testing.assert_array_equal(masks, ref["mask"])
    sizes = np.array([np.load(tmp_path / ("sizes_%d.npy" % r)) for r in range(world)])
    assert sizes[:, 0].sum() == world * n_local          # every point is owned by exactly one slab
    assert (np.abs(sizes[:, 0] - n_local) < 0.05 * n_local).all()   # equal-count slabs
    assert (sizes[:, 1] > 0).all() and sizes[:, 1].sum() < 0.9 * world * n_local   # halos are a fraction, not a replica
    for r in range(world):
        st = np.load(tmp_path / ("stats_%d.npy" % r))
        for got, key in zip(st, ("mean", "std", "threshold")):
            assert np.float32(got).tobytes() == np.float32(ref[key]).tobytes(), (r, key)


def _unequal_worker(rank, world, port, sizes, k, sigma, mode, out_dir):
    """shards of different sizes: `mode` slab = dist_slab.slab_sor, replicated = dist.sharded_sor (the fallback)"""
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import datasets
    from oracle.slab_backend import GlooHostComm, NumpySlabBackend
    full = datasets.uniform(sum(sizes), 10.0, 42)
    lo = sum(sizes[:rank])
    mine = full[lo:lo + sizes[rank]]
    be = NumpySlabBackend()
    if mode == "slab":
        slab = importlib.import_module("3dgsconverter_amd.dist_slab")
        try:
            res = slab.slab_sor(be, GlooHostComm(be), be.rows_buffer(mine), sizes[rank], k, sigma, want_host=True)
            np.save(os.path.join(out_dir, "mask_%d.npy" % rank), res["mask_host"])
            np.save(os.path.join(out_dir, "md_%d.npy" % rank), res["mean_dists_host"])
            np.save(os.path.join(out_dir, "stats_%d.npy" % rank), res["stats_host"])
        except slab.SlabUnsupported as e:
            with open(os.path.join(out_dir, "unsupported_%d.txt" % rank), "w") as f:
                f.write(str(e))
    else:
        gdist = importlib.import_module("3dgsconverter_amd.dist")
        rows = be.rows_buffer(mine) if len(mine) else be.buf("rows_empty", 16)
        if mode == "either":   # dist.sharded_sor: the slab exchange, else (declined on every rank together) the replicated one
            res, path = gdist.sharded_sor(be, GlooHostComm(be), rows, sizes[rank], k, sigma, want_host=True)
            open(os.path.join(out_dir, "path_%d_%s" % (rank, path)), "w").close()
        else:
            res = gdist.replicated_sor(be, GlooHostComm(be), rows, sizes[rank], k, sigma, want_host=True)
        np.save(os.path.join(out_dir, "mask_%d.npy" % rank), res["mask_host"])
        np.save(os.path.join(out_dir, "md_%d.npy" % rank), res["mean_dists_host"])
        np.save(os.path.join(out_dir, "stats_%d.npy" % rank), res["stats_host"])
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,sizes", [("slab", (16384, 9001)), ("slab", (12000, 8192, 20003)), ("replicated", (7001, 3000)),
                                        ("replicated", (5000, 0, 6001)), ("either", (9000, 3000)), ("either", (9000, 9500))])
def test_unequal_index_shards_equal_single_process(mode, sizes, tmp_path):
    """round 3: shards of different sizes (what a per-rank density filter leaves behind), starts that are not multiples of
    4 (unaligned piece sums) -- mean distances, statistics and masks == the un-sharded oracle, bit for bit; the replicated
    exchange also takes an EMPTY shard"""
    import torch.multiprocessing as mp
    from oracle import datasets, sor as osor
    world, k, sigma = len(sizes), 16, 1.0
    mp.spawn(_unequal_worker, args=(world, _free_port(), list(sizes), k, sigma, mode, str(tmp_path)), nprocs=world, join=True)
    assert not list(tmp_path.glob("unsupported_*"))
    if mode == "either":   # a shard below 8192 points: every rank falls back together; otherwise the slabs serve it
        want = "replicated" if min(sizes) < 8192 else "slab"
        assert len(list(tmp_path.glob("path_*_" + want))) == world
    ref = osor.sor(datasets.uniform(sum(sizes), 10.0, 42), k, sigma, workers=2)
    md = np.concatenate([np.load(tmp_path / ("md_%d.npy" % r)) for r in range(world)])
    np.testing.assert_array_equal(md.view(np.uint32), ref["mean_dists"].view(np.uint32))
    masks = np.concatenate([np.load(tmp_path / ("mask_%d.npy" % r)) for r in range(world)]).astype(bool)
    np.testing.assert_array_equal(masks, ref["mask"])
    for r in range(world):
        st = np.load(tmp_path / ("stats_%d.npy" % r))
        assert np.float32(st[2]).tobytes() == np.float32(ref["threshold"]).tobytes()


def test_slab_sor_declines_small_shards_on_every_rank(tmp_path):
    """a shard below numpy's 8192-element piece: SlabUnsupported (a SlabUncertain) from EVERY rank in the same step --
    decided from the all-gathered histograms, so nobody is left waiting in a collective"""
    import importlib
    import torch.multiprocessing as mp
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    assert issubclass(slab.SlabUnsupported, slab.SlabUncertain)
    sizes = [9000, 5000, 0]
    mp.spawn(_unequal_worker, args=(3, _free_port(), sizes, 8, 1.0, "slab", str(tmp_path)), nprocs=3, join=True)
    assert len(list(tmp_path.glob("unsupported_*"))) == 3 and not list(tmp_path.glob("mask_*"))


# ---------------------------------------------------------------------------------------------------------------
# multi-GPU density (3dgsconverter_amd/dist_density.py): per-rank voxel histograms -> merged counts -> host clusters ->
# per-rank mask; then BASELINE.json configs[2] sharded: density -> (unequal survivors per rank) -> slab SOR
def _density_worker(rank, world, port, sizes, spec, kwargs, then_sor, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    dd = importlib.import_module("3dgsconverter_amd.dist_density")
    from oracle import datasets
    from oracle.slab_backend import GlooHostComm, NumpySlabBackend
    full = datasets.make(spec)
    lo = sum(sizes[:rank])
    mine = full[lo:lo + sizes[rank]]
    be = NumpySlabBackend()
    comm = GlooHostComm(be)
    rows = be.rows_buffer(mine) if len(mine) else be.buf("rows_empty", 16)
    res = dd.sharded_density(be, comm, rows, sizes[rank], **kwargs)
    mask = np.zeros(sizes[rank], bool) if res["empty"] else be.to_host(res["mask"], np.uint8, sizes[rank]).astype(bool)
    np.save(os.path.join(out_dir, "dmask_%d.npy" % rank), mask)
    with open(os.path.join(out_dir, "dinfo_%d.json" % rank), "w") as f:
        import json
        json.dump({k: res[k] for k in ("n_total", "min_points", "n_unique", "kept_clusters", "max_len", "kept", "empty")}, f)
    if then_sor and not res["empty"]:
        n2 = res["kept"]
        rows2, orig2 = be.buf("rows2", 12 * max(n2, 1)), be.buf("orig2", 4 * max(n2, 1))
        assert be.compact_rows(rows, res["mask"], sizes[rank], rows2, orig2) == n2
        r2 = slab.slab_sor(be, comm, rows2, n2, then_sor[0], then_sor[1], want_host=True)
        final = np.zeros(sizes[rank], bool)
        final[be.to_host(orig2, np.uint32, n2)[r2["mask_host"]]] = True
        np.save(os.path.join(out_dir, "final_%d.npy" % rank), final)
        np.save(os.path.join(out_dir, "stats_%d.npy" % rank), r2["stats_host"])
    dist.destroy_process_group()


@pytest.mark.parametrize("sizes,spec,kwargs", [
    ((30000, 20001), {"kind": "uniform", "n": 50001, "extent": 5.0, "seed": 3}, {"sensitivity": 0.5}),
    ((9000, 0, 21000), {"kind": "two_blobs", "n": 30000, "seed": 2}, {"voxel_size": 0.5, "threshold_percentage": 0.05, "keep_multicluster": True}),
    ((15000, 15000), {"kind": "clustered", "n": 30000, "seed": 5}, {"voxel_size": 1.0, "threshold_percentage": 0.32}),
    ((10000, 10000), {"kind": "uniform", "n": 20000, "extent": 10.0, "seed": 4}, {"sensitivity": 0.5}),        # removes everything
    # a tiny shard next to a large one: the large rank's unique-voxel count exceeds the tiny rank's list capacity, so its
    # list buffers are re-allocated AFTER the histogram was written (contents must survive the growth)
    ((20, 29980), {"kind": "uniform", "n": 30000, "extent": 10.0, "seed": 6}, {"voxel_size": 0.5, "threshold_percentage": 0.01}),
])
def test_sharded_density_equals_the_reference_filter(sizes, spec, kwargs, tmp_path):
    """masks of the index shards (unequal, one EMPTY) == oracle/density.py (pinned to the reference's own
    apply_density_filter) on the whole cloud; min_points comes from the GLOBAL size (data_processor.py:48)"""
    import json
    import torch.multiprocessing as mp
    from oracle import datasets, density as oden
    world = len(sizes)
    mp.spawn(_density_worker, args=(world, _free_port(), list(sizes), spec, kwargs, None, str(tmp_path)), nprocs=world, join=True)
    full = datasets.make(spec)
    kw = dict(kwargs)
    if "sensitivity" in kw:
        kw["voxel_size"], kw["threshold_percentage"] = oden.params_from_sensitivity(kw.pop("sensitivity"))
    ref = oden.density_filter(full, **kw)
    got = np.concatenate([np.load(tmp_path / ("dmask_%d.npy" % r)) for r in range(world)])
    np.testing.assert_array_equal(got, ref["mask"])
    for r in range(world):
        info = json.load(open(tmp_path / ("dinfo_%d.json" % r)))
        assert info["n_total"] == len(full) and info["min_points"] == ref["min_points"] and info["n_unique"] == ref["unique_voxels"]
        assert info["empty"] == (not ref["mask"].any())


def test_sharded_density_then_slab_sor_is_the_reference_chain(tmp_path):
    """BASELINE.json configs[2] sharded: density (sensitivity 0.5) leaves a different number of survivors on every rank; the
    slab SOR of those unequal shards == the reference chain on the whole cloud (oracle density -> oracle SOR)"""
    import torch.multiprocessing as mp
    from oracle import datasets, density as oden, sor as osor
    spec = {"kind": "uniform", "n": 60000, "extent": 5.0, "seed": 9}
    sizes = [32000, 28000]
    mp.spawn(_density_worker, args=(2, _free_port(), sizes, spec, {"sensitivity": 0.5}, (16, 1.0), str(tmp_path)), nprocs=2, join=True)
    full = datasets.make(spec)
    v, t = oden.params_from_sensitivity(0.5)
    dref = oden.density_filter(full, v, t)
    assert 0 < dref["mask"].sum() < len(full)
    sref = osor.sor(full[dref["mask"]], 16, 1.0, workers=2)
    want = np.zeros(len(full), bool)
    want[np.nonzero(dref["mask"])[0][sref["mask"]]] = True
    got = np.concatenate([np.load(tmp_path / ("final_%d.npy" % r)) for r in range(2)])
    np.testing.assert_array_equal(got, want)
    for r in range(2):
        assert np.float32(np.load(tmp_path / ("stats_%d.npy" % r))[2]).tobytes() == np.float32(sref["threshold"]).tobytes()


def test_slab_sor_refuses_what_it_cannot_certify(tmp_path):
    """far floaters have their neighbours beyond any halo: every rank raises SlabUncertain (the caller then uses the
    replicated exchange, which is exact for any cloud) -- never a silently wrong result"""
    import torch.multiprocessing as mp
    world, n_local = 2, 10000
    mp.spawn(_slab_worker, args=(world, _free_port(), n_local, 16, 1.0, "floaters", str(tmp_path)), nprocs=world, join=True)
    assert len(list(tmp_path.glob("uncertain_*"))) == world and not list(tmp_path.glob("mask_*"))


def test_slab_plan_is_equal_count_and_monotone():
    import importlib
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    rng = np.random.default_rng(0)
    hist = rng.integers(0, 50, 4096)
    hist[1000:1010] = 5000  # a dense sheet
    for world in (1, 2, 3, 8):
        cut, total = slab.plan_slabs(hist, world)
        assert cut[0] == 0 and cut[-1] == 4096 and len(cut) == world + 1 and total == hist.sum()
        assert all(a <= b for a, b in zip(cut, cut[1:]))
        own = [hist[a:b].sum() for a, b in zip(cut, cut[1:])]
        assert max(own) <= total / world + 5000 + 50
    assert slab.pts_per_cell(16) == pytest.approx(7.25) and slab.pts_per_cell(32) == pytest.approx(14.5)


# ---------------------------------------------------------------------------------------------------------------
# SOG SH-palette K-Means: the reference's independent chunks (sog.py:527-552) dealt out to the ranks
def _oracle_kmeans(data, k, it, init):
    from oracle import kmeans as okm
    if k >= len(data):
        return data.copy(), np.arange(len(data), dtype=np.int32)
    c, l, _ = okm.lloyd(data, init, it)
    return c, l


def _palette_worker(rank, world, port, n, d, level, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    pal = importlib.import_module("3dgsconverter_amd.dist_palette")
    from oracle.slab_backend import GlooHostComm, NumpySlabBackend
    be = NumpySlabBackend()
    sh = (np.random.default_rng(5).standard_normal((n, d)) * 0.1).astype(np.float32)
    np.random.seed(77)
    cen, lab = pal.palette_kmeans(sh, level, 4, kmeans=_oracle_kmeans, comm=GlooHostComm(be), be=be)
    np.save(os.path.join(out_dir, "cen_%d.npy" % rank), cen)
    np.save(os.path.join(out_dir, "lab_%d.npy" % rank), lab)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_palette_chunks_across_ranks_equal_the_single_process_loop(world, tmp_path):
    import importlib
    import torch.multiprocessing as mp
    pal = importlib.import_module("3dgsconverter_amd.dist_palette")
    n, d, level = 7000, 9, 8   # 6 chunks of 1167 rows, k_per_chunk 683
    mp.spawn(_palette_worker, args=(world, _free_port(), n, d, level, str(tmp_path)), nprocs=world, join=True)
    sh = (np.random.default_rng(5).standard_normal((n, d)) * 0.1).astype(np.float32)
    np.random.seed(77)
    cen, lab = pal.palette_kmeans(sh, level, 4, kmeans=_oracle_kmeans)
    plan = pal.palette_plan(n, level)
    assert plan == {"target_k": 4096, "num_chunks": 6, "chunk_size": 1167, "k_per_chunk": 683}
    assert cen.shape == (6 * 683, d) and lab.shape == (n,) and lab.max() < len(cen)
    for r in range(world):
        np.testing.assert_array_equal(np.load(tmp_path / ("cen_%d.npy" % r)), cen)
        np.testing.assert_array_equal(np.load(tmp_path / ("lab_%d.npy" % r)), lab)


# ---------------------------------------------------------------------------------------------------------------
# the step's plan: C (gsx_slab_plan, what gsx_sor_slab_step_dev uses) == Python (dist_slab.plan_step, what the CPU
# choreography above uses), field by field; and the launcher plumbing (3dgsconverter_amd/launch.py)
def test_c_plan_equals_python_plan():
    import ctypes as C
    import importlib
    L = importlib.import_module("3dgsconverter_amd._lib")
    S = importlib.import_module("3dgsconverter_amd.dist_slab")
    lib = L.load()
    rng = np.random.default_rng(0)
    statuses = set()
    for trial in range(120):
        G = int(rng.integers(1, 9))
        kind = trial % 5
        words = np.zeros(8 + 4096 * G, np.uint32)
        box = np.zeros(8, np.float32)
        mn = rng.normal(size=3).astype(np.float32) * 10
        ext = rng.random(3).astype(np.float32) * np.float32([5, 50, 1][trial % 3])
        if kind == 4:
            ext[1] = 0
        box[:3], box[3:6] = -mn, mn + ext
        if trial == 7:
            box[4] = np.inf
        words[:8] = box.view(np.uint32)
        h = rng.integers(0, 50, (G, 4096)).astype(np.uint32)
        if kind == 1:
            h[:, 1000:1010] += 5000            # a dense sheet
        if kind == 2:
            h[:] = 0
            h[:, 100:140] = rng.integers(0, 3000, (G, 40))   # everything inside 40 bins: halos cover most of it
        if kind == 3:
            h = (h * rng.integers(0, 2, (G, 1))).astype(np.uint32)   # empty shards
        words[8:] = h.reshape(-1)
        k = int(rng.choice([8, 16, 25, 32]))
        for r in range(G):
            n_local = int(h[r].sum())
            plan = L.SlabPlan()
            assert lib.gsx_slab_plan(words.ctypes.data, G, r, n_local, k, 1.5, C.byref(plan)) == 0, L.last_error()
            c, p = plan.as_dict(), S.plan_step(words, G, r, n_local, k, 1.5)
            assert p["status"] == c["status"]
            statuses.add(p["status"])
            if p["status"] == 0:
                for key, a in p.items():
                    same = (np.float32(a).tobytes() == np.float32(c[key]).tobytes()) if isinstance(a, float) else a == c[key]
                    assert same, (trial, G, r, key, a, c[key])
                assert sum(p["own_cnt"]) == n_local and p["n_send"] == n_local + sum(p["halo_cnt"])
    assert statuses == {S.SLAB_OK, S.SLAB_EMPTY, S.SLAB_NONFINITE, S.SLAB_SMALL_SHARD, S.SLAB_NO_STRUCTURE}


def _rdzv_rank(rank, path, out):
    sys.path.insert(0, ROOT)
    import importlib
    launch = importlib.import_module("3dgsconverter_amd.launch")
    uid = launch.exchange_unique_id(rank, lambda: bytes(range(128)), path, timeout_s=30)
    with open(out + ".%d" % rank, "wb") as f:
        f.write(uid)


def _agree_rank(rank, world, uid, path, out):
    sys.path.insert(0, ROOT)
    import importlib
    os.environ.pop("GSX_COMM_TRANSPORT", None)
    launch = importlib.import_module("3dgsconverter_amd.launch")
    with open(out + ".%d" % rank, "w") as f:
        f.write(launch.agree_transport(rank, world, uid, path, timeout_s=30))


@pytest.mark.parametrize("uids,expect", [(["0000:05:00.0", "0000:15:00.0", "0000:25:00.0"], "rccl"),
                                         (["0000:05:00.0", "0000:15:00.0", "0000:05:00.0"], "hostwire")])
def test_transport_follows_the_devices_the_ranks_opened(tmp_path, uids, expect):
    """ranks on distinct GPUs (whatever each one's HIP_VISIBLE_DEVICES shows) -> RCCL; any two on one GPU -> hostwire"""
    import importlib
    import multiprocessing as mp
    path, out = str(tmp_path / "uid"), str(tmp_path / "got")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_agree_rank, args=(r, 3, uids[r], path, out)) for r in (2, 0, 1)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert [p.exitcode for p in procs] == [0, 0, 0]
    assert [open(out + ".%d" % r).read() for r in range(3)] == [expect] * 3
    os.environ["WORLD_SIZE"] = "3"
    try:
        importlib.import_module("3dgsconverter_amd.launch").retire_unique_id(0, path)
    finally:
        os.environ.pop("WORLD_SIZE", None)
    assert not [f for f in os.listdir(tmp_path) if f.startswith("uid")]


def test_unique_id_rendezvous_through_a_file(tmp_path):
    import importlib
    import multiprocessing as mp
    launch = importlib.import_module("3dgsconverter_amd.launch")
    path, out = str(tmp_path / "uid"), str(tmp_path / "got")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rdzv_rank, args=(r, path, out)) for r in (2, 1, 0)]   # rank 0 last: the others wait for it
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
    assert [p.exitcode for p in procs] == [0, 0, 0]
    for r in range(3):
        assert open(out + ".%d" % r, "rb").read() == bytes(range(128))
    launch.retire_unique_id(0, path)
    assert not os.path.exists(path)
    assert launch.pick_device_and_transport(3, 8, 8)[:2] == (3, os.environ.get("GSX_COMM_TRANSPORT") or "rccl")
    assert launch.pick_device_and_transport(1, 2, 1)[:2] == (0, os.environ.get("GSX_COMM_TRANSPORT") or "hostwire")


def test_comm_watchdog_ends_a_stuck_job_with_a_json_line():
    """launch.comm_watchdog (no GPU needed): a stage that does not finish within the timeout -> rank 0 writes one JSON line
    with an "error" field and the process exits 124; stage() restarts the clock, done() disarms it"""
    import json
    import subprocess
    import textwrap
    code = textwrap.dedent("""
        import importlib, time
        launch = importlib.import_module("3dgsconverter_amd.launch")
        w = launch.comm_watchdog(0, 2, 1, 1.0, {"metric": "m", "n_gpus": 2})
        w.stage("first")
        time.sleep(0.6)
        w.stage("second")          # the clock restarts: 0.6 + 0.6 s pass without a timeout
        time.sleep(0.6)
        w.stage("third")
        time.sleep(30)
        print("not reached")
    """)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=60)
    assert r.returncode == 124
    d = json.loads(r.stdout.strip())
    assert d["value"] is None and d["n_gpus"] == 2 and "'third'" in d["error"]
    code2 = textwrap.dedent("""
        import importlib, time
        launch = importlib.import_module("3dgsconverter_amd.launch")
        w = launch.comm_watchdog(1, 2, 1, 0.3, {})
        w.done()
        time.sleep(1.0)
        print("finished")
    """)
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, cwd=ROOT, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "finished"
