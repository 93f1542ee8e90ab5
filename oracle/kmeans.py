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


def assign_threaded(data: np.ndarray, centroids: np.ndarray, threads: int | None = None) -> np.ndarray:
    """``assign`` on row slices in parallel (the C call releases the GIL): the scalar restatement of gpu_ops.py:57-73 at
    BASELINE configs[4]'s own shape (156 250 x 45, K = 1024: 7e9 multiply-adds per assign) in seconds instead of a minute"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    data = np.ascontiguousarray(data, dtype=np.float32)
    cent = np.ascontiguousarray(centroids, dtype=np.float32)
    n, d = data.shape
    threads = max(1, min(threads or (os.cpu_count() or 1), 64, (n + 1023) // 1024))
    labels = np.zeros(n, dtype=np.int32)
    lib = clib()
    cuts = [n * t // threads for t in range(threads + 1)]

    def run(t):
        a, b = cuts[t], cuts[t + 1]
        if b > a:
            lib.gsxo_kmeans_assign(data[a:b].ctypes.data, b - a, d, cent.ctypes.data, cent.shape[0], labels[a:b].ctypes.data)

    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(run, range(threads)))
    return labels


def update_f64(data: np.ndarray, labels: np.ndarray, k: int):
    """gpu_ops.py:75-96 with double accumulation (order-insensitive) -> (centroids f32[k,d], counts)"""
    data = np.ascontiguousarray(data, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    cent = np.zeros((k, data.shape[1]), dtype=np.float32)
    counts = np.zeros(k, dtype=np.int32)
    clib().gsxo_kmeans_update(data.ctypes.data, data.shape[0], data.shape[1], labels.ctypes.data, k, cent.ctypes.data, counts.ctypes.data)
    return cent, counts


def near_tie_tolerance(d: int) -> float:
    """relative gap below which two f32 squared distances over d dimensions cannot be ordered reliably: each is a sum of
    d non-negative products accumulated in binary32 (error <= (d + 1) 2^-24 relative, with or without FMA contraction --
    the reference's Taichi backend may contract); two of them: twice that, plus SURVEY's 4 ulp"""
    return 2.0 * (d + 1) * 2.0 ** -24 + 4 * 2.0 ** -23


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


# ---- round 3: the reference's scikit-learn path (gpu_ops.py:48-52, sog.py:561) and the device's scalar solver --------
def sklearn_minibatch(data: np.ndarray, k: int, max_iter: int = 10):
    """processing/gpu_ops.py:48-52 verbatim: the reference's CPU path (unseeded; k-means++ seeding inside sklearn)"""
    from sklearn.cluster import MiniBatchKMeans
    N = len(data)
    km = MiniBatchKMeans(n_clusters=k, max_iter=max_iter, batch_size=min(4096 * 4, N), n_init="auto", compute_labels=True)
    km.fit(data)
    return km.cluster_centers_.astype(np.float32), km.labels_.astype(np.int32)


def sklearn_codebook_561(flat: np.ndarray, k: int = 256):
    """formats/sog.py:561-562 verbatim: MiniBatchKMeans(n_clusters=256, n_init='auto').fit(centroids_flat); sorted"""
    from sklearn.cluster import MiniBatchKMeans
    km = MiniBatchKMeans(n_clusters=k, n_init="auto").fit(np.asarray(flat, dtype=np.float32).reshape(-1, 1))
    return np.array(sorted(km.cluster_centers_.flatten()), dtype=np.float32)


def inertia_1d(vals: np.ndarray, centroids: np.ndarray) -> float:
    """sum of squared distances to the nearest centroid (float64)"""
    c = np.sort(np.asarray(centroids, dtype=np.float64).reshape(-1))
    x = np.asarray(vals, dtype=np.float64).reshape(-1)
    i = np.clip(np.searchsorted(c, x), 0, len(c) - 1)
    left = np.maximum(i - 1, 0)
    d = np.minimum(np.abs(x - c[i]), np.abs(x - c[left]))
    return float((d * d).sum())


K1_BINS = 1024


def _k1_start(xs: np.ndarray, k: int, mode: int) -> np.ndarray:
    """companded start of csrc/kmeans1d.hip:k1_lloyd_kernel: mode 0 = count^(1/3) over uniform bins of [min, max],
    mode 1 = width^(2/3) over equal-count bins"""
    n = len(xs)
    lo, hi = float(xs[0]), float(xs[-1])
    b = np.arange(K1_BINS + 1)
    if mode == 0:
        edges = lo + (hi - lo) * (b / K1_BINS)
        edges[-1] = hi
        pos = np.searchsorted(xs.astype(np.float64), edges, side="right")
        pos[0], pos[-1] = 0, n
        w = np.cbrt(np.diff(pos).astype(np.float64))
    else:
        edges = xs[(b * (n - 1)) // K1_BINS].astype(np.float64)
        width = np.diff(edges)
        w = np.cbrt(width * width)
    W = np.concatenate([[0.0], np.cumsum(w)])
    if not W[-1] > 0:
        return np.full(k, lo, dtype=np.float32)
    t = (np.arange(k) + 0.5) / k * W[-1]
    b0 = np.clip(np.searchsorted(W, t, side="right") - 1, 0, K1_BINS - 1)
    wb = W[b0 + 1] - W[b0]
    frac = np.where(wb > 0, (t - W[b0]) / np.where(wb > 0, wb, 1.0), 0.5)
    return (edges[b0] + frac * (edges[b0 + 1] - edges[b0])).astype(np.float32)


def kmeans1d_sorted(vals: np.ndarray, k: int, iters: int = 50):
    """numpy restatement of csrc/kmeans1d.hip (the device's replacement for the scikit-learn scalar codebooks): sort,
    float64 prefix sums, Lloyd on run boundaries (ties to the lower centroid) from both companded starts, lower
    inertia wins.  -> (centroids f32[k] ascending, inertia[3] = chosen, start 0, start 1)"""
    xs = np.sort(np.asarray(vals, dtype=np.float32).reshape(-1))
    n = len(xs)
    x64 = xs.astype(np.float64)
    p1 = np.concatenate([[0.0], np.cumsum(x64)])
    p2 = np.concatenate([[0.0], np.cumsum(x64 * x64)])
    best, out = None, [0.0, 0.0, 0.0]
    for mode in (0, 1):
        c = _k1_start(xs, k, mode)
        for it in range(iters + 1):
            mid = (c[:-1].astype(np.float64) + c[1:].astype(np.float64)) * 0.5
            b = np.concatenate([[0], np.searchsorted(x64, mid, side="right"), [n]])
            b = np.maximum.accumulate(b)
            cnt = np.diff(b)
            s1 = p1[b[1:]] - p1[b[:-1]]
            if it == iters:
                s2 = p2[b[1:]] - p2[b[:-1]]
                cd = c.astype(np.float64)
                inertia = float(np.where(cnt > 0, s2 - 2.0 * cd * s1 + cnt * cd * cd, 0.0).sum())
                break
            c = np.where(cnt > 0, s1 / np.maximum(cnt, 1), c).astype(np.float32)
        out[1 + mode] = inertia
        if best is None or inertia < best[0]:
            best = (inertia, c)
    out[0] = best[0]
    return best[1], np.array(out)


def kmeans_pp_restated(data: np.ndarray, k: int, uniforms: np.ndarray, n_local_trials: int | None = None) -> np.ndarray:
    """csrc/kmeans_pp.hip == sklearn's _kmeans_plusplus (greedy): row floor(u0 n), then for every further centroid
    n_local_trials candidates -- the rows where cumsum(min squared distance) first exceeds u * total -- of which the one
    with the smallest potential sum_i min(mind_i, d_i) wins (first minimum).  float64 arithmetic (the device uses float32
    fma chains: picks agree except where a cumulative sum is within rounding of its target).  -> chosen row indices"""
    x = np.asarray(data, dtype=np.float64)
    n = len(x)
    L = 2 + int(np.log(k)) if n_local_trials is None else int(n_local_trials)
    u = np.asarray(uniforms, dtype=np.float64)
    assert u.shape == (1 + (k - 1) * L,)
    idx = [min(int(u[0] * n), n - 1)]
    mind = ((x - x[idx[0]]) ** 2).sum(1)
    for t in range(1, k):
        ut = u[1 + (t - 1) * L: 1 + t * L]
        cs = np.cumsum(mind)
        if not cs[-1] > 0:
            cand = np.minimum((ut * n).astype(np.int64), n - 1)
        else:
            cand = np.minimum(np.searchsorted(cs, ut * cs[-1], side="right"), n - 1)
        dist = ((x[None, :, :] - x[cand][:, None, :]) ** 2).sum(2)
        pots = np.minimum(mind[None, :], dist).sum(1)
        b = int(np.argmin(pots))
        idx.append(int(cand[b]))
        mind = np.minimum(mind, dist[b])
    return np.asarray(idx)
