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
