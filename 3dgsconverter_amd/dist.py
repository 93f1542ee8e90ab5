"""Multi-GPU SOR: one process per GPU, RCCL over xGMI.

The reference is single-process; this is the MI355X-native scale-out of
its SOR path.  Rank r holds a contiguous index range of the cloud (ranges in rank
order, any sizes) in its own HBM as (n_local,3) rows (how a loader would hand out a file), and gets the
survivor mask of exactly that range back.

One step:
  1. all-gather of the xyz rows -> every GPU holds the full reference set (the only data every
     query needs); torch.distributed backend "nccl" IS RCCL on ROCm;
  2. every GPU bins the full set (identical grid on every rank) and computes exact KNN mean
     distances for its SHARE OF THE GRID'S BRICKS -- a spatial slab, gsx_sor_knn_share_dev --
     writing them at their original indices of a zero-filled n_total array.  (Sharing out the
     QUERIES BY INDEX instead leaves every brick with 1/world of its lanes live: measured on one
     GPU, 1/8 of the queries of an 8M cloud cost as much as all of them.)
  3. sum all-reduce of that array: every entry has exactly one non-zero contributor, so the
     sum is exact and every rank now holds the f32 mean distances of the whole cloud.  NOT an
     all-reduce of partial STATISTICS: the reference's threshold is numpy's pairwise f32
     mean/std over the whole array, whose rounding depends on the global element order
     (8192-element pieces), so every rank evaluates it redundantly and bit-exactly;
  4. mask of the local index range against the (identical on every rank) threshold.

torch is only plumbing here (device memory, streams, the process group).  The compute
callables are injectable so that the choreography can be exercised on CPU with the gloo
backend (tests/test_dist_cpu.py plugs the oracle in; nothing in the product does).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional


PARALLELISM = ("index-sharded input; all-gather of xyz rows, per-rank slab of the grid's bricks, sum all-reduce of "
               "the mean distances (RCCL), statistics evaluated redundantly per rank")


@dataclass
class ShardedSorResult:
    mask_local: "object"        # uint8/bool tensor [n_local]
    mean_dists_local: "object"  # f32 tensor [n_local]
    stats: "object"             # f32 tensor [3]: mean, std, threshold (identical on every rank)
    n_total: int


class HipCompute:
    """Default compute backend: the C ABI on the current torch device / stream."""

    def __init__(self, device_index: int = 0):
        import torch
        from . import _lib
        self.torch = torch
        self._lib = _lib
        self.ctx = _lib.Context(device_index)
        self.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        self.adaptive = False

    def set_adaptive(self, on: bool):
        """adaptive KNN grid for the calls below: exact and fast on clouds with far floaters -- what the
        replicated exchange is the fallback for -- at the price of one host synchronisation per call"""
        self.adaptive = bool(on)
        self.ctx.set_param("adaptive", 1 if on else 0)

    def check(self):
        """synchronise and raise GsxError if the asynchronous KNN calls met non-finite coordinates (their outputs were
        set to NaN, so the statistics are NaN and the mask all-false)"""
        self.ctx.check()

    def knn(self, xyz_all, q_begin: int, q_count: int, k: int, algo: int = 0):
        t = self.torch
        assert xyz_all.is_contiguous() and xyz_all.dtype == t.float32 and xyz_all.shape[1] == 3
        out = t.empty(q_count, dtype=t.float32, device=xyz_all.device)
        base = xyz_all.data_ptr()
        self.ctx.sor_knn(base, base + 4, base + 8, 3, xyz_all.shape[0], q_begin, q_count, k, out.data_ptr(), algo=algo)
        return out

    def knn_share(self, xyz_all, k: int, share: int, nshares: int, algo: int = 0):
        """f32[n_total]: this share's mean distances at their original indices, +0.0 elsewhere."""
        t = self.torch
        assert xyz_all.is_contiguous() and xyz_all.dtype == t.float32 and xyz_all.shape[1] == 3
        out = t.empty(xyz_all.shape[0], dtype=t.float32, device=xyz_all.device)
        base = xyz_all.data_ptr()
        self.ctx.sor_knn_share(base, base + 4, base + 8, 3, xyz_all.shape[0], k, share, nshares, out.data_ptr(), algo=algo)
        return out

    def stats(self, md_all, threshold_factor: float):
        t = self.torch
        st = t.empty(4, dtype=t.float32, device=md_all.device)
        self.ctx.sor_stats(md_all.data_ptr(), md_all.numel(), threshold_factor, st.data_ptr())
        return st[:3]

    def mask(self, md_local, stats):
        t = self.torch
        out = t.empty(md_local.numel(), dtype=t.uint8, device=md_local.device)
        self.ctx.sor_mask(md_local.data_ptr(), md_local.numel(), stats.data_ptr() + 8, out.data_ptr())
        return out


def sharded_sor(xyz_local, k: int, threshold_factor: float, compute, group=None, algo: int = 0) -> ShardedSorResult:
    """xyz_local: (n_local,3) float32 tensor: index shard `rank` of the cloud (consecutive index ranges in rank order; the
    shards may have different sizes, also zero -- they are padded to the longest for the all-gather)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_local = xyz_local.shape[0]
    start = 0
    if world > 1:
        sizes = torch.zeros(world, dtype=torch.int64, device=xyz_local.device)
        sizes[rank] = n_local
        dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
        sizes = [int(v) for v in sizes.tolist()]
        nmax, n_total, start = max(sizes), sum(sizes), sum(sizes[:rank])
        if n_total == 0:
            raise ValueError("sor: empty cloud")
        if min(sizes) == nmax:
            xyz_all = torch.empty((world * n_local, 3), dtype=xyz_local.dtype, device=xyz_local.device)
            dist.all_gather_into_tensor(xyz_all, xyz_local.contiguous(), group=group)
        else:
            padded = torch.zeros((nmax, 3), dtype=xyz_local.dtype, device=xyz_local.device)
            padded[:n_local] = xyz_local
            gathered = torch.empty((world * nmax, 3), dtype=xyz_local.dtype, device=xyz_local.device)
            dist.all_gather_into_tensor(gathered, padded, group=group)
            xyz_all = torch.cat([gathered[q * nmax:q * nmax + sizes[q]] for q in range(world)]).contiguous()
    else:
        n_total = n_local
        xyz_all = xyz_local.contiguous()
    if world > 1:
        md_all = compute.knn_share(xyz_all, k, rank, world, algo)
        dist.all_reduce(md_all, op=dist.ReduceOp.SUM, group=group)
        md_local = md_all[start:start + n_local].clone()  # own, aligned storage for the mask kernel
    else:
        md_all = md_local = compute.knn(xyz_all, 0, n_local, k, algo)
    stats = compute.stats(md_all, threshold_factor)
    mask = compute.mask(md_local, stats) if n_local else torch.zeros(0, dtype=torch.uint8, device=xyz_local.device)
    return ShardedSorResult(mask, md_local, stats, n_total)
