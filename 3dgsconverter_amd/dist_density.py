"""Multi-GPU voxel-density filter: per-rank voxel histograms -> merged counts.

The reference is single-process (data_processor.py:11-117).  Rank r holds an index shard of the cloud (any size, also 0)
and gets the survivor mask of that shard back -- identical to the single-GPU / reference result:

  1. n_total = sum all-reduce of the shard sizes: ``min_points = int(n_total * thr / 100)`` is over the GLOBAL cloud
     (data_processor.py:48);
  2. ``gsx_density_hist_dev``: voxel occupancy of the local shard (the single-GPU hash-table kernels), every occupied voxel
     exported as an absolute int64 key triple + int64 count in device memory;
  3. the list lengths, then the padded lists, are all-gathered (RCCL from the C library; 32 B per occupied voxel -- at
     the reference's voxel sizes a few hundred to a few hundred thousand entries, never the points);
  4. ``gsx_density_merge_dev`` on every rank: counts of equal keys added in a hash table, voxels with
     ``count >= min_points`` back in np.unique row order -- what ``gsx_density_voxels`` returns for the whole cloud;
  5. the 6-connected clusters and the keep rule on the host (processing/clusters.py, <= ~1000 dense voxels, identical on
     every rank), ``gsx_density_mask_dev`` on the local rows.
No collective inside kernels; one all-reduce + two all-gathers per call.  ``Comm`` / backend are injectable as in
dist_slab.py (tests run the same choreography on CPU with gloo and the numpy backend of oracle/slab_backend.py).
"""
from __future__ import annotations

import numpy as np

from .dist_slab import KIND_I64_SUM
from .processing import clusters as _clusters
from .processing.data_processor import density_params_from_sensitivity


def sharded_density(be, comm, rows, n_local: int, voxel_size=1.0, threshold_percentage=0.32, sensitivity=None,
                    keep_multicluster=False):
    """rows: backend buffer with this rank's (n_local,3) float32 rows.
    -> dict(mask: backend buffer u8[n_local] (None when nothing survives anywhere), kept: survivors of this shard,
            n_total, min_points, n_unique, kept_clusters, max_len, empty: True when the filter removed every point)"""
    if sensitivity is not None:
        voxel_size, threshold_percentage = density_params_from_sensitivity(sensitivity)
    G = comm.world
    n_local = int(n_local)
    # ---- 1. global size
    cnt = be.buf("dd_cnt", 16)
    be.from_host(cnt, np.array([n_local], dtype=np.int64))
    if G > 1:
        comm.all_reduce(cnt, 1, KIND_I64_SUM)
    n_total = int(be.to_host(cnt, np.int64, 1)[0])
    min_points = int(n_total * (threshold_percentage / 100.0))      # data_processor.py:48
    out = {"n_total": n_total, "min_points": min_points, "mask": None, "kept": 0, "empty": True, "n_unique": 0,
           "kept_clusters": 0, "max_len": 0}
    if n_total == 0:
        return out
    # ---- 2. local histogram (device lists)
    cap = max(n_local, 1)
    keys = be.buf("dd_keys", 24 * cap)
    counts = be.buf("dd_counts", 8 * cap)
    u_local = be.density_hist(rows, n_local, float(voxel_size), cap, keys, counts) if n_local else 0
    # ---- 3. all-gather: lengths, then the lists padded to the longest
    if G > 1:
        lens_s, lens_r = be.buf("dd_len_s", 8), be.buf("dd_len_r", 8 * G)
        be.from_host(lens_s, np.array([u_local], dtype=np.int64))
        comm.all_gather(lens_s, lens_r, 8)
        lens = be.to_host(lens_r, np.int64, G).astype(np.int64)
        umax = int(lens.max())
        if umax == 0:
            return out
        gk, gc = be.buf("dd_gkeys", 24 * umax * G), be.buf("dd_gcounts", 8 * umax * G)
        keys = be.buf("dd_keys", 24 * umax, keep=24 * u_local)       # (a reallocation keeps the first u_local entries:
        counts = be.buf("dd_counts", 8 * umax, keep=8 * u_local)     #  a tiny shard next to a large one)
        if u_local < umax:                        # padding entries carry count 0: the merge ignores them, keys included
            be.pad_density_list(keys, counts, u_local, umax)
        comm.all_gather(keys, gk, 24 * umax)
        comm.all_gather(counts, gc, 8 * umax)
        m = umax * G
    else:
        gk, gc, m = keys, counts, u_local
    # ---- 4. merge (identical on every rank)
    dense_cap = int(min(n_total, n_total // max(min_points, 1) + 1))
    occ = be.density_merge(gk, gc, m, min_points, dense_cap)
    out["n_unique"] = occ["n_unique"]
    if len(occ["dense_keys"]) == 0:
        return out
    # ---- 5. clusters on the host, mask on the local rows
    comps = _clusters.connected_clusters(map(tuple, occ["dense_keys"].tolist()))
    kept, kept_clusters, max_len = _clusters.select_clusters(comps, keep_multicluster)
    out["kept_clusters"], out["max_len"] = kept_clusters, max_len
    if not kept:
        return out
    kept_keys = np.array(sorted(kept), dtype=np.int64).reshape(-1, 3)
    mask = be.buf("dd_mask", n_local + 16)
    if n_local:
        be.density_mask(rows, n_local, float(voxel_size), kept_keys, mask)
        out["kept"] = int(be.to_host(mask, np.uint8, n_local).astype(bool).sum())
    out["mask"], out["empty"] = mask, False
    return out
