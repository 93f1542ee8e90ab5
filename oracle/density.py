"""Voxel-density / connected-cluster filter -- CPU restatement of the reference.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/gsconverter/processing/data_processor.py:11-117
(``DataProcessor.apply_density_filter``).  Pinned against the reference's own
function run in the build container (oracle/make_golden.py -> tests/golden/).
"""
from __future__ import annotations

from collections import deque

import numpy as np


def params_from_sensitivity(sensitivity: float):
    """data_processor.py:24-28."""
    voxel_size = 2.0 - (sensitivity * 1.8)
    voxel_size = max(0.1, voxel_size)
    threshold_percentage = 0.1 + (sensitivity * 0.9)
    return voxel_size, threshold_percentage


def voxel_keys(xyz: np.ndarray, voxel_size: float) -> np.ndarray:
    """data_processor.py:38-39: f32 coords / python float (weak => f32 divide), floor, int64."""
    coords = np.ascontiguousarray(xyz, dtype=np.float32)
    return np.floor(coords / voxel_size).astype(np.int64)


def min_points_for(n: int, threshold_percentage: float) -> int:
    """data_processor.py:48."""
    return int(n * (threshold_percentage / 100.0))


def cluster_dense_voxels(dense_voxels: np.ndarray, keep_multicluster: bool):
    """6-connected components over the dense voxels and the keep rule.

    data_processor.py:57-106.  ``dense_voxels`` must be in np.unique(axis=0)
    (lexicographic) order: the reference builds a python set from tuples in that
    order and iterates the set, so the cluster discovery order -- which breaks
    ties between equally large clusters through the stable sort at line 95 --
    is the set's iteration order.
    Returns (valid_voxel_set, kept_clusters, max_len).
    """
    dense_set = set(map(tuple, dense_voxels.tolist()))
    visited = set()
    clusters = []
    for voxel in dense_set:
        if voxel in visited:
            continue
        comp = {voxel}
        visited.add(voxel)
        queue = deque([voxel])
        while queue:
            cx, cy, cz = queue.popleft()
            for nb in ((cx - 1, cy, cz), (cx + 1, cy, cz), (cx, cy - 1, cz),
                       (cx, cy + 1, cz), (cx, cy, cz - 1), (cx, cy, cz + 1)):
                if nb in dense_set and nb not in visited:
                    visited.add(nb)
                    comp.add(nb)
                    queue.append(nb)
        clusters.append(comp)
    if not clusters:
        return set(), 0, 0
    clusters.sort(key=len, reverse=True)
    max_len = len(clusters[0])
    min_cluster_size = max_len * 0.05 if keep_multicluster else max_len
    valid = set()
    kept = 0
    for c in clusters:
        if len(c) >= min_cluster_size:
            valid.update(c)
            kept += 1
            if not keep_multicluster:
                break
    return valid, kept, max_len


def density_filter(xyz: np.ndarray, voxel_size: float = 1.0, threshold_percentage: float = 0.32,
                   sensitivity=None, keep_multicluster: bool = False):
    """-> dict(mask bool[N], unique_voxels, min_points, kept_clusters, max_len, voxel_size, threshold_percentage)."""
    if sensitivity is not None:
        voxel_size, threshold_percentage = params_from_sensitivity(sensitivity)
    n = len(xyz)
    keys = voxel_keys(xyz, voxel_size)
    uniq, inverse, counts = np.unique(keys, axis=0, return_inverse=True, return_counts=True)
    inverse = np.asarray(inverse).reshape(-1)
    min_points = min_points_for(n, threshold_percentage)
    dense_idx = np.where(counts >= min_points)[0]
    info = {"unique_voxels": len(uniq), "min_points": min_points, "voxel_size": voxel_size,
            "threshold_percentage": threshold_percentage, "kept_clusters": 0, "max_len": 0}
    if len(dense_idx) == 0:
        info["mask"] = np.zeros(n, dtype=bool)
        return info
    valid, kept, max_len = cluster_dense_voxels(uniq[dense_idx], keep_multicluster)
    in_cluster = np.zeros(len(uniq), dtype=bool)
    for j in dense_idx:  # only dense voxels can be valid (line 111 tests every voxel)
        if tuple(uniq[j].tolist()) in valid:
            in_cluster[j] = True
    info.update(mask=in_cluster[inverse], kept_clusters=kept, max_len=max_len)
    return info
