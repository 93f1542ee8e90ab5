"""K-Means codebook (SOG writer) -- CPU restatement of the reference.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PINNED (round 2): the reference's K-Means kernels are Taichi
(/root/reference/gsconverter/processing/gpu_ops.py:57-96, driver :178-191) and Taichi is not
installable here, but their bodies are plain Python: oracle/taichi_shim.py executes them
with Taichi's default types, ``np.random.seed`` pins the unseeded ``np.random.choice`` of
gpu_ops.py:182, and oracle/make_golden_kmeans.py stores what the reference's own code
returned (tests/golden/kmeans_ref.*).  ``lloyd(..., accumulate="f32seq")`` below reproduces
those fixtures bit for bit (tests/test_kmeans_ref_golden.py); ``accumulate="f64"`` is the
order-insensitive variant the GPU result is compared with inside SURVEY.md 8(c)'s tolerance.
``quantize_to_codebook`` and ``sog_sh_plan`` restate closures of ``SogFormat.write``
(formats/sog.py:408-419, 513-529) and are pinned by decoding the bundle the reference wrote
and by a spy on its ``gpu_ops.kmeans`` calls (same generator).
"""
from __future__ import annotations

import numpy as np

from . import clib


def lloyd(data: np.ndarray, init_centroids: np.ndarray, max_iter: int = 10, accumulate: str = "f64"):
    """gpu_ops.py:178-191 with ``centroids_np`` injected instead of np.random.choice.

    Exactly ``max_iter`` x (assign, update); no convergence test; returns the
    post-update centroids with the pre-update labels (one step stale), as the
    reference does.  Returns (centroids f32[K,D], labels i32[N], counts i32[K]).
    """
    data = np.ascontiguousarray(data, dtype=np.float32)
    n, d = data.shape
    cent = np.ascontiguousarray(init_centroids, dtype=np.float32).copy()
    k = cent.shape[0]
    labels = np.zeros(n, dtype=np.int32)
    counts = np.zeros(k, dtype=np.int32)
    lib = clib()
    # "f64": sums in double, rounded once (the order-independent centre of what the reference's f32
    # atomics can produce); "f32seq": binary32 accumulation in index order -- what the reference's kernel
    # computes when its parallel loop runs sequentially (oracle/taichi_shim.py), bit for bit.
    update = {"f64": lib.gsxo_kmeans_update, "f32seq": lib.gsxo_kmeans_update_f32seq}[accumulate]
    for _ in range(max_iter):
        lib.gsxo_kmeans_assign(data.ctypes.data, n, d, cent.ctypes.data, k, labels.ctypes.data)
        update(data.ctypes.data, n, d, labels.ctypes.data, k, cent.ctypes.data, counts.ctypes.data)
    return cent, labels, counts


def assign(data: np.ndarray, centroids: np.ndarray) -> np.ndarray:
    """gpu_ops.py:57-73 (k_means_assign) on its own."""
    data = np.ascontiguousarray(data, dtype=np.float32)
    cent = np.ascontiguousarray(centroids, dtype=np.float32)
    labels = np.zeros(len(data), dtype=np.int32)
    clib().gsxo_kmeans_assign(data.ctypes.data, data.shape[0], data.shape[1], cent.ctypes.data,
                              cent.shape[0], labels.ctypes.data)
    return labels


def assign_margin(data: np.ndarray, centroids: np.ndarray, labels_a: np.ndarray, labels_b: np.ndarray):
    """For points where two label vectors disagree: relative gap between the two f32 distances.
    Used for the 'agree unless the two smallest distances are within a few ulp' rule."""
    data = np.asarray(data, dtype=np.float64)
    cent = np.asarray(centroids, dtype=np.float64)
    idx = np.nonzero(labels_a != labels_b)[0]
    da = ((data[idx] - cent[labels_a[idx]]) ** 2).sum(1)
    db = ((data[idx] - cent[labels_b[idx]]) ** 2).sum(1)
    return idx, np.abs(da - db) / np.maximum(np.maximum(da, db), 1e-300)


def inertia(data: np.ndarray, centroids: np.ndarray, labels: np.ndarray) -> float:
    data = np.asarray(data, dtype=np.float64)
    cent = np.asarray(centroids, dtype=np.float64)
    return float(((data - cent[labels]) ** 2).sum())


def kmeans_front_door_shortcut(data: np.ndarray, k: int):
    """gpu_ops.py:30-31: k >= N returns (data.copy(), arange(N, int32))."""
    n = data.shape[0]
    if k >= n:
        return data.copy(), np.arange(n, dtype=np.int32)
    return None


def quantize_to_codebook(vals: np.ndarray, cb: np.ndarray) -> np.ndarray:
    """formats/sog.py:408-419: nearest entry of a sorted codebook, ties -> right neighbour."""
    if len(cb) == 1:
        return np.zeros_like(vals, dtype=np.uint8)
    idx = np.searchsorted(cb, vals)
    idx = np.clip(idx, 0, len(cb) - 1)
    left = np.maximum(idx - 1, 0)
    d_idx = np.abs(vals - cb[idx])
    d_left = np.abs(vals - cb[left])
    use_left = d_left < d_idx
    idx[use_left] = left[use_left]
    return idx.astype(np.uint8)


def sog_sh_plan(n: int, compression_level: int = 0):
    """formats/sog.py:513-529: palette size and chunking of the SH-N K-Means."""
    official_standard_k = min(64, 2 ** int(np.floor(np.log2(n / 1024)))) * 1024
    if compression_level <= 3:
        target_k = min(65536, official_standard_k)
    elif compression_level <= 6:
        target_k = min(16384, official_standard_k)
    else:
        target_k = min(4096, official_standard_k)
    target_k = max(256, target_k)
    num_chunks = max(1, min(64, n // 1024))
    chunk_size = int(np.ceil(n / num_chunks))
    k_per_chunk = max(16, int(np.ceil(target_k / num_chunks)))
    return {"target_k": int(target_k), "num_chunks": int(num_chunks), "chunk_size": chunk_size,
            "k_per_chunk": k_per_chunk}
