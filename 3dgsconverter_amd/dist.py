"""Multi-GPU SOR: one process per GPU, splats sharded BY INDEX, RCCL over xGMI.

The reference is single-process; this is the MI355X-native scale-out of
its SOR path.  Rank r owns the contiguous index range
[r*n_local, (r+1)*n_local) of the cloud, resident in its own HBM as (n_local,3) rows.

One step:
  1. all-gather of the xyz rows  -> every GPU holds the full reference set (the only data
     every query needs); torch.distributed backend "nccl" IS RCCL on ROCm;
  2. each GPU bins the full set and computes exact KNN mean distances for ITS queries only
     (gsx_sor_knn_dev with q_begin/q_count) -- no collective inside the kernel path;
  3. all-gather of the f32 mean distances (4 B/splat).  NOT an all-reduce of partial
     sums: the reference's threshold is numpy's pairwise f32 mean/std over the WHOLE array,
     whose rounding depends on the global element order (8192-element pieces), so every
     rank evaluates the statistics redundantly and bit-exactly on the gathered array;
  4. mask of the local shard against the (identical on every rank) threshold.

torch is only plumbing here (device memory, streams, the process group).  The compute
callables are injectable so that the choreography can be exercised on CPU with the gloo
backend (tests/test_dist_cpu.py plugs the oracle in; nothing in the product does).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional


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

    def knn(self, xyz_all, q_begin: int, q_count: int, k: int, algo: int = 0):
        t = self.torch
        assert xyz_all.is_contiguous() and xyz_all.dtype == t.float32 and xyz_all.shape[1] == 3
        out = t.empty(q_count, dtype=t.float32, device=xyz_all.device)
        base = xyz_all.data_ptr()
        self.ctx.sor_knn(base, base + 4, base + 8, 3, xyz_all.shape[0], q_begin, q_count, k, out.data_ptr(), algo=algo)
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


_equal_shards_checked = set()  # (group id, n_local): the size check costs two collectives + a host sync, do it once


def sharded_sor(xyz_local, k: int, threshold_factor: float, compute, group=None, algo: int = 0) -> ShardedSorResult:
    """xyz_local: (n_local,3) float32 tensor, same n_local on every rank (index shard `rank`)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n_local = xyz_local.shape[0]
    if world > 1:
        key = (id(group), n_local, world)
        if key not in _equal_shards_checked:
            sizes = torch.tensor([n_local], dtype=torch.int64, device=xyz_local.device)
            lo, hi = sizes.clone(), sizes.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
            if int(lo) != int(hi):
                raise ValueError("sharded_sor needs equally sized index shards (got %d..%d)" % (int(lo), int(hi)))
            _equal_shards_checked.add(key)
        xyz_all = torch.empty((world * n_local, 3), dtype=xyz_local.dtype, device=xyz_local.device)
        dist.all_gather_into_tensor(xyz_all, xyz_local.contiguous(), group=group)
    else:
        xyz_all = xyz_local.contiguous()
    md_local = compute.knn(xyz_all, rank * n_local, n_local, k, algo)
    if world > 1:
        md_all = torch.empty(world * n_local, dtype=md_local.dtype, device=md_local.device)
        dist.all_gather_into_tensor(md_all, md_local, group=group)
    else:
        md_all = md_local
    stats = compute.stats(md_all, threshold_factor)
    mask = compute.mask(md_local, stats)
    return ShardedSorResult(mask, md_local, stats, world * n_local)
