"""-m gpu: the multi-GPU slab path (3dgsconverter_amd/dist_slab.py) with the REAL device entry points on the one GPU
a test box has:

  * world 1 through RcclComm -- librccl is dlopen'ed, a one-rank communicator is created and every collective of the
    step (all-reduce, all-gather, grouped send/recv) goes through RCCL on the library's stream;
  * world 2 and 3 as separate processes that share the GPU, collectives through gloo on staged host copies
    (TorchHostComm): the partition with halos, the slab KNN with reference-only rows, the certificate, the return
    path and the piece-sum statistics run in HIP, only the wire is emulated.  The 8-GPU run over xGMI is the driver's.

Bar: identical to the single-GPU result, which is itself pinned to the reference's golden vectors."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import sha16
from oracle import datasets, sor as osor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_slab_path_world1_through_rccl_matches_the_golden_1m(gsx, golden_cases):
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    case = golden_cases["sor"]["sor_u1m_k16_s1"]
    xyz = datasets.make(case["dataset"])
    be = slab.HipSlabBackend(0)
    comm = slab.RcclComm(be.ctx, 0, 1, slab.RcclComm.unique_id())
    rows = be.buf("rows", xyz.nbytes)
    be.from_host(rows, xyz)
    res = slab.slab_sor(be, comm, rows, len(xyz), case["k_used"], case["sigma_used"], want_host=True)
    assert sha16(res["mean_dists_host"].tobytes()) == case["mean_dists_sha"]
    assert np.float32(res["stats_host"][2]).tobytes().hex() == case["threshold_hex"]
    assert sha16(np.packbits(res["mask_host"]).tobytes()) == case["mask_sha"]
    assert res["n_own"] == len(xyz) and res["n_halo"] == 0
    # the RCCL wrappers themselves, on known data
    a = be.buf("t_a", 64)
    be.from_host(a, np.arange(8, dtype=np.float32))
    comm.all_reduce(a, 8, slab.KIND_F32_MAX)
    np.testing.assert_array_equal(be.to_host(a, np.float32, 8), np.arange(8, dtype=np.float32))
    b = be.buf("t_b", 64)
    comm.all_gather(a, b, 32)
    np.testing.assert_array_equal(be.to_host(b, np.float32, 8), np.arange(8, dtype=np.float32))
    comm.all_to_all_v(a, [2], [3], b, [5], [3], 4)
    np.testing.assert_array_equal(be.to_host(b, np.float32, 8)[5:8], [2, 3, 4])
    be.check()
    comm.close()


def _spawn(target, world, args):
    """world processes through the standard library's spawn context.  NOT torch.multiprocessing: importing torch into
    this (pytest) process after libgsx_hip.so has bound ROCm's own libamdhip64 / librccl puts two HIP runtimes into
    one process, which corrupts the heap at exit (torch must be imported FIRST where both are used --
    the workers do that)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world) + tuple(args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert [p.exitcode for p in procs] == [0] * world, [p.exitcode for p in procs]


def _worker(rank, world, port, n_local, k, sigma, kind, out_dir):
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401  (first: see _spawn)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    slab = importlib.import_module("3dgsconverter_amd.dist_slab")
    full = _cloud(kind, world * n_local)
    be = slab.HipSlabBackend(0)
    comm = slab.TorchHostComm(be)
    rows = be.buf("rows", 12 * n_local)
    be.from_host(rows, full[rank * n_local:(rank + 1) * n_local])
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


def _cloud(kind, n):
    if kind == "uniform":
        return datasets.uniform(n, 10.0, 42)
    if kind == "gradient":
        xyz = datasets.uniform(n, 1.0, 7)
        xyz[:, 1] = (xyz[:, 1] ** 2) * np.float32(40.0)
        xyz[:, 0] *= np.float32(6.0)
        return xyz
    return datasets.scene_with_floaters(n, 3)


@pytest.mark.parametrize("world,kind,n_local,k", [(2, "uniform", 300000, 16), (3, "uniform", 100000, 32), (2, "gradient", 150000, 16),
                                                  (3, "gradient", 70000, 8)])
def test_slab_path_on_shared_gpu_equals_the_unsharded_result(world, kind, n_local, k, tmp_path):
    _spawn(_worker, world, (_free_port(), n_local, k, 1.0, kind, str(tmp_path)))
    assert not list(tmp_path.glob("uncertain_*"))
    full = _cloud(kind, world * n_local)
    ref = osor.sor(full, k, 1.0)
    md = np.concatenate([np.load(tmp_path / ("md_%d.npy" % r)) for r in range(world)])
    bad = np.nonzero(md.view(np.uint32) != ref["mean_dists"].view(np.uint32))[0]
    assert len(bad) == 0, (len(bad), bad[:5], md[bad[:5]], ref["mean_dists"][bad[:5]])
    masks = np.concatenate([np.load(tmp_path / ("mask_%d.npy" % r)) for r in range(world)]).astype(bool)
    np.testing.assert_array_equal(masks, ref["mask"])
    for r in range(world):
        st = np.load(tmp_path / ("stats_%d.npy" % r))
        for got, key in zip(st, ("mean", "std", "threshold")):
            assert np.float32(got).tobytes() == np.float32(ref[key]).tobytes(), (r, key)
    sizes = np.array([np.load(tmp_path / ("sizes_%d.npy" % r)) for r in range(world)])
    assert sizes[:, 0].sum() == world * n_local and (sizes[:, 1] > 0).all()
    assert sizes[:, 1].sum() < 0.6 * world * n_local


def test_slab_path_refuses_floaters_on_every_rank(tmp_path):
    _spawn(_worker, 2, (_free_port(), 40000, 16, 1.0, "floaters", str(tmp_path)))
    assert len(list(tmp_path.glob("uncertain_*"))) == 2 and not list(tmp_path.glob("mask_*"))
