"""Statistical Outlier Removal -- CPU restatement of the reference path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows /root/reference/gsconverter/processing/data_processor.py:119-182
(``DataProcessor.remove_flyers``, CPU branch) and the identical threshold code
of gpu_ops.py:259-263.  The exact-KNN path (scipy cKDTree) is the authoritative
one; the reference never applies the mask it computes
(data_processor.py:180-182) -- the mask of line 180 is what is
returned here.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import clib


def synth_xyz(n: int, extent: float = 10.0, seed: int = 0) -> np.ndarray:
    """Synthetic cloud of SURVEY.md section 8(c): uniform f32 in [0, extent)^3."""
    rng = np.random.default_rng(seed)
    return rng.random((n, 3), dtype=np.float32) * np.float32(extent)


def params_from_intensity(intensity: float):
    """data_processor.py:125-134 -- intensity in [1,10] -> (k, threshold_factor)."""
    k = int(10 + (intensity - 1) * (40 / 9))
    threshold_factor = 20.0 - (intensity - 1) * (17.0 / 9)
    return k, threshold_factor


def mean_dists_ckdtree(xyz: np.ndarray, k: int, chunk_size: int = 50000, workers: int | None = None) -> np.ndarray:
    """data_processor.py:156-173: cKDTree build, chunked query(k+1), row mean, f32 store."""
    from scipy.spatial import cKDTree

    coords = np.ascontiguousarray(xyz, dtype=np.float32)
    n = len(coords)
    if workers is None:
        workers = max(1, (os.cpu_count() or 2) - 1)  # cpu_count()-1, data_processor.py:171
    tree = cKDTree(coords)
    out = np.zeros(n, dtype=np.float32)
    for i in range(0, n, chunk_size):
        end = min(i + chunk_size, n)
        dists, _ = tree.query(coords[i:end], k=k + 1, workers=workers)
        out[i:end] = np.mean(dists[:, 1:], axis=1)  # exclude self at column 0
    return out


def threshold_numpy(mean_dists: np.ndarray, threshold_factor: float):
    """data_processor.py:176-178 executed by numpy itself -> (mean, std, threshold) as np.float32."""
    global_mean = np.mean(mean_dists)
    global_std = np.std(mean_dists)
    threshold = global_mean + threshold_factor * global_std
    return global_mean, global_std, threshold


def threshold_c(mean_dists: np.ndarray, threshold_factor: float):
    """Explicit restatement of the numpy arithmetic (oracle/gsx_oracle.c gsxo_sor_stats_f32)."""
    md = np.ascontiguousarray(mean_dists, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    clib().gsxo_sor_stats_f32(md.ctypes.data, md.size, float(threshold_factor), out.ctypes.data)
    return out[0], out[1], out[2]


def sor(xyz: np.ndarray, k: int = 25, threshold_factor: float = 10.5, intensity=None, workers=None):
    """Full reference CPU path -> dict(mean_dists f32[N], mean, std, threshold, mask bool[N])."""
    if intensity is not None:
        k, threshold_factor = params_from_intensity(intensity)
    md = mean_dists_ckdtree(xyz, k, workers=workers)
    m, s, t = threshold_numpy(md, threshold_factor)
    with np.errstate(invalid="ignore"):
        mask = md < t  # strict, data_processor.py:180
    return {"mean_dists": md, "mean": m, "std": s, "threshold": t, "mask": mask, "k": k,
            "threshold_factor": threshold_factor}


def mean_dists_brute_c(xyz: np.ndarray, k: int) -> np.ndarray:
    """O(N^2) scalar C restatement of the cKDTree arithmetic (no scipy)."""
    coords = np.ascontiguousarray(xyz, dtype=np.float32)
    out = np.zeros(len(coords), dtype=np.float32)
    clib().gsxo_sor_mean_dists_brute(coords.ctypes.data, len(coords), int(k), out.ctypes.data)
    return out


def mean_dists_brute_subset_c(xyz: np.ndarray, k: int, qidx: np.ndarray) -> np.ndarray:
    coords = np.ascontiguousarray(xyz, dtype=np.float32)
    q = np.ascontiguousarray(qidx, dtype=np.int64)
    out = np.zeros(len(q), dtype=np.float32)
    clib().gsxo_sor_mean_dists_brute_subset(coords.ctypes.data, len(coords), int(k), q.ctypes.data, len(q),
                                            out.ctypes.data)
    return out


def pairwise_sum_f32(a: np.ndarray) -> np.float32:
    a = np.ascontiguousarray(a, dtype=np.float32)
    return np.float32(clib().gsxo_pairwise_sum_f32(a.ctypes.data, a.size))


def mean_dists_brute_numpy(xyz: np.ndarray, k: int) -> np.ndarray:
    """Pure-numpy O(N^2) restatement (tiny N only): f64 (dx*dx+dy*dy)+dz*dz, k+1 smallest,
    sqrt, np.mean of columns 1..k, f32 store."""
    c = np.ascontiguousarray(xyz, dtype=np.float32).astype(np.float64)
    d = c[:, None, :] - c[None, :, :]
    s = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    n = len(c)
    kk = k + 1
    if n < kk:
        s = np.concatenate([s, np.full((n, kk - n), np.inf)], axis=1)
    part = np.sort(s, axis=1)[:, :kk]
    dist = np.sqrt(part)
    return np.mean(dist[:, 1:], axis=1).astype(np.float32)
