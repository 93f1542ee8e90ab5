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


class OracleCompute:
    def knn(self, xyz_all, q_begin, q_count, k, algo=0):
        import torch
        from oracle import sor as osor
        md = osor.mean_dists_ckdtree(xyz_all.numpy(), k, workers=2)
        return torch.from_numpy(md[q_begin:q_begin + q_count].copy())

    def knn_share(self, xyz_all, k, share, nshares, algo=0):
        # any partition of the queries works for the choreography; like the GPU one this is
        # spatial (slabs along z), i.e. scattered in index space
        import torch
        from oracle import sor as osor
        xyz = xyz_all.numpy()
        md = osor.mean_dists_ckdtree(xyz, k, workers=2)
        order = np.argsort(xyz[:, 2], kind="stable")
        n = len(xyz)
        mine = order[n * share // nshares: n * (share + 1) // nshares]
        out = np.zeros(n, np.float32)
        out[mine] = md[mine]
        return torch.from_numpy(out)

    def stats(self, md_all, factor):
        import torch
        from oracle import sor as osor
        return torch.tensor([np.float32(v) for v in osor.threshold_numpy(md_all.numpy(), factor)], dtype=torch.float32)

    def mask(self, md_local, stats):
        return (md_local < stats[2]).to(dtype=__import__("torch").uint8)


def _worker(rank, world, port, n_local, k, sigma, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gdist = importlib.import_module("3dgsconverter_amd.dist")
    from oracle import datasets
    full = datasets.uniform(world * n_local, 10.0, 42)
    local = torch.from_numpy(full[rank * n_local:(rank + 1) * n_local].copy())
    res = gdist.sharded_sor(local, k, sigma, OracleCompute())
    np.save(os.path.join(out_dir, "mask_%d.npy" % rank), res.mask_local.numpy())
    np.save(os.path.join(out_dir, "stats_%d.npy" % rank), res.stats.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sor_equals_single_process(world, tmp_path):
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
    import torch
    from oracle import datasets, sor as osor
    gdist = importlib.import_module("3dgsconverter_amd.dist")
    xyz = datasets.uniform(3000, 10.0, 1)
    res = gdist.sharded_sor(torch.from_numpy(xyz), 8, 1.0, OracleCompute())
    np.testing.assert_array_equal(res.mask_local.numpy().astype(bool), osor.sor(xyz, 8, 1.0, workers=2)["mask"])


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
    from oracle.slab_backend import NumpySlabBackend
    full = _slab_cloud(kind, world * n_local)
    be = NumpySlabBackend()
    comm = slab.TorchHostComm(be)
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
    from oracle.slab_backend import NumpySlabBackend
    be = NumpySlabBackend()
    sh = (np.random.default_rng(5).standard_normal((n, d)) * 0.1).astype(np.float32)
    np.random.seed(77)
    cen, lab = pal.palette_kmeans(sh, level, 4, kmeans=_oracle_kmeans, comm=slab.TorchHostComm(be), be=be)
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
