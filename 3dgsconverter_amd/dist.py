"""Multi-GPU SOR, the replicated exchange: north_star's "all-gather of reference tiles" design, RCCL from the C library.

The reference is single-process; its SOR treats queries as independent units over one reference
set (data_processor.py:167-173) and thresholds on numpy's f32 mean/std of the whole mean-distance array (:176-180).
Rank r holds a contiguous index range of the cloud (ranges in rank order, ANY sizes, also zero) in its own HBM as
(n_local,3) rows, and gets the survivor mask of exactly that range back -- bit-identical to the single-GPU result.

One step (``replicated_sor``), every collective a ``gsx_comm_*`` call on the context's stream (csrc/comm.hip):
  1. all-gather of the shard sizes (8 bytes per rank; the step's host synchronisation), then of the xyz rows, padded to the
     longest shard -> every GPU holds the full reference set (the only data every query needs);
  2. every GPU bins the full set (identical grid on every rank) and computes exact KNN mean distances for its SHARE OF THE
     GRID'S BRICKS -- a spatial slab, ``gsx_sor_knn_share_dev`` -- writing them at their original indices of a zero-filled
     n_total array.  (Sharing out the QUERIES BY INDEX instead leaves every brick with 1/world of its lanes live: measured on
     one GPU, 1/8 of the queries of an 8M cloud cost as much as all of them.)
  3. float32 sum all-reduce of that array: every entry has exactly one non-zero contributor, so the sum is exact and every
     rank holds the mean distances of the whole cloud.  NOT an all-reduce of partial STATISTICS: the reference's threshold is
     numpy's pairwise f32 mean/std over the whole array, whose rounding depends on the global element order (8192-element
     pieces), so every rank evaluates it redundantly and bit-exactly;
  4. mask of the local index range against the (identical on every rank) threshold.

Cost per rank: 12 (G-1) N_local bytes received and G-fold redundant binning -- which is why the slab exchange
(dist_slab.py) is the default and this one its fallback: it is exact for ANY cloud (far floaters, blobs, tiny or empty
shards), where the slab exchange declines (``SlabUncertain``).  ``sharded_sor`` tries one, then the other.

numpy + ctypes only: no torch anywhere on the multi-GPU path.  ``Comm`` / backend are injectable as in dist_slab.py
(tests/test_dist_cpu.py runs the same choreography on CPU with gloo and the numpy backend of oracle/slab_backend.py).
"""
from __future__ import annotations

import numpy as np

from .dist_slab import KIND_F32_SUM, SlabUncertain, slab_sor

PARALLELISM = ("index-sharded input; all-gather of xyz rows, per-rank slab of the grid's bricks, sum all-reduce of "
               "the mean distances (RCCL from libgsx_hip.so), statistics evaluated redundantly per rank")


class ReplicatedResult(dict):
    """mask / mean_dists / stats: backend buffers of the LOCAL index range (stats: mean, std, threshold of the whole cloud)"""

    def check(self):
        self["_be"].check()
        return self


def replicated_sor(be, comm, rows, n_local: int, k: int, threshold_factor: float, algo: int = 0, want_host: bool = False):
    """rows: backend buffer with this rank's (n_local,3) float32 index shard; comm: None = one rank"""
    G = comm.world if comm is not None else 1
    r = comm.rank if comm is not None else 0
    n_local = int(n_local)
    # ---- 1. sizes, then the rows (padded to the longest shard for the all-gather)
    if G > 1:
        sz_s, sz_r = be.buf("rep_size_s", 8), be.buf("rep_size_r", 8 * G)
        be.from_host(sz_s, np.array([n_local], dtype=np.int64))
        comm.all_gather(sz_s, sz_r, 8)
        sizes = [int(v) for v in be.to_host(sz_r, np.int64, G)]                # <- the step's host synchronisation
    else:
        sizes = [n_local]
    nmax, n_total, start = max(sizes), sum(sizes), sum(sizes[:r])
    if n_total == 0:
        raise ValueError("sor: empty cloud")
    if G > 1:
        xyz_all = be.buf("rep_xyz", 12 * n_total)
        if min(sizes) == nmax:
            comm.all_gather(rows, xyz_all, 12 * n_local)
        else:
            padded = rows
            if n_local < nmax:   # (the padding rows are never looked at: only the first sizes[q] rows of a block are kept)
                padded = be.buf("rep_pad", 12 * nmax)
                if n_local:
                    be.copy(padded, rows, 12 * n_local)
            gathered = be.buf("rep_gather", 12 * nmax * G)
            comm.all_gather(padded, gathered, 12 * nmax)
            o = 0
            for q in range(G):
                if sizes[q]:
                    be.copy(be.at(xyz_all, 12 * o), be.at(gathered, 12 * nmax * q), 12 * sizes[q])
                o += sizes[q]
    else:
        xyz_all = rows
    # ---- 2./3. this rank's share of the bricks, sum all-reduce
    md_all = be.buf("rep_md", 4 * (n_total + 4))
    if G > 1:
        be.knn_share(xyz_all, n_total, k, r, G, md_all, algo)
        comm.all_reduce(md_all, n_total, KIND_F32_SUM)
    else:
        be.knn_all(xyz_all, n_total, k, md_all, algo)
    # ---- 4. statistics of the whole array (redundantly, numpy-exact), mask of the local range
    stats = be.buf("rep_stats", 16)
    be.stats(md_all, n_total, threshold_factor, stats)
    md = be.at(md_all, 4 * start)
    if n_local and (4 * start) % 16:   # the mask kernel reads 16 bytes at a time
        md = be.buf("rep_md_local", 4 * (n_local + 4))
        be.copy(md, be.at(md_all, 4 * start), 4 * n_local)
    mask = be.buf("rep_mask", n_local + 4)
    if n_local:
        be.mask(md, n_local, stats, mask)
    out = ReplicatedResult({"mask": mask, "mean_dists": md, "stats": stats, "_be": be, "n_total": n_total, "sizes": sizes})
    if want_host:
        out.check()
        out["mask_host"] = be.to_host(mask, np.uint8, n_local).view(np.bool_)
        out["mean_dists_host"] = be.to_host(md, np.float32, n_local)
        out["stats_host"] = be.to_host(stats, np.float32, 3)
    return out


def sharded_sor(be, comm, rows, n_local: int, k: int, threshold_factor: float, want_host: bool = False):
    """The multi-GPU SOR entry point: the slab exchange, and -- when it declines on every rank together (its certificate
    failed, shards too small, no slab structure) -- the replicated exchange, which is exact for any cloud.
    -> (result, path) with path in {"slab", "replicated"}"""
    try:
        res = slab_sor(be, comm, rows, n_local, k, threshold_factor)
        res.check()    # SlabUncertain is raised on every rank in the same step (the count is all-reduced)
        if want_host:
            res["mask_host"] = be.to_host(res["mask"], np.uint8, n_local).view(np.bool_)
            res["mean_dists_host"] = be.to_host(res["mean_dists"], np.float32, n_local)
            res["stats_host"] = be.to_host(res["stats"], np.float32, 3)
        return res, "slab"
    except SlabUncertain:
        pass
    if hasattr(be, "set_adaptive"):
        be.set_adaptive(True)   # the clouds that get here are the ones the adaptive paths exist for
    return replicated_sor(be, comm, rows, n_local, k, threshold_factor, want_host=want_host), "replicated"
